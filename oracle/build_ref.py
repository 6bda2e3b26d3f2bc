"""Build the reference's own CPU operators into oracle/_ref/ (test infrastructure; authoring container only).

    python oracle/build_ref.py        ->  oracle/_ref/disprcnn_ref_cpu*.so

The reference's ROIAlign / NMS CPU kernels are two plain C++ files (csrc/cpu/ROIAlign_cpu.cpp, csrc/cpu/nms_cpu.cpp) that
need nothing but the torch headers of this image.  They are compiled from where they lie under /root/reference; the only
change is the token patch SURVEY 8c records -- torch >= 1.5 removed the `Tensor::type()` overload of AT_DISPATCH_*:

    ROIAlign_cpu.cpp:242   AT_DISPATCH_FLOATING_TYPES(input.type(), ...  ->  input.scalar_type()
    nms_cpu.cpp:71         AT_DISPATCH_FLOATING_TYPES(dets.type(),  ...  ->  dets.scalar_type()

The patch is applied to a scratch copy under a temporary directory (never inside the repo); only the compiled module lands in
oracle/_ref/ (git-ignored; it travels to the GPU box with the snapshot).  No stand-in headers, no generated code.
/root/reference absent (GPU box) => nothing to do, the prebuilt module -- if present -- is used as is.
"""
import glob
import os
import re
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CSRC = "/root/reference/disprcnn/csrc"
OUT = os.path.join(HERE, "_ref")
NAME = "disprcnn_ref_cpu"

# The three reference files that get compiled / included are pinned by content: anything else than the surveyed sources is refused
# before a compiler or an import sees it (build() runs this from the driver's test process).
SHA256 = {
    "cpu/ROIAlign_cpu.cpp": "826804743beed9b01bea0b823939eeb96fd61a93fef9144c1fb2bb11c53f3e84",
    "cpu/nms_cpu.cpp": "38d0949a259c5c86db73d7c46f9222f29bd252fc18cd0d6a12f09e0a233724c3",
    "cpu/vision.h": "2f677f95be56f4a9621e75e9febb608b5a93462f75059838d7a8efd7a8164020",
}

PATCHES = {   # file -> [(line number, old token, new token)]
    "cpu/ROIAlign_cpu.cpp": [(242, "input.type()", "input.scalar_type()")],
    "cpu/nms_cpu.cpp": [(71, "dets.type()", "dets.scalar_type()")],
}


def built_module_path():
    hits = sorted(glob.glob(os.path.join(OUT, NAME + "*.so")))
    return hits[0] if hits else None


def load():
    """Import the prebuilt module (None if it was never built)."""
    path = built_module_path()
    if path is None:
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build(force=False, verbose=False):
    if not os.path.isdir(REF_CSRC):
        return built_module_path()
    if built_module_path() and not force:
        return built_module_path()
    import hashlib
    for rel, want in SHA256.items():
        got = hashlib.sha256(open(os.path.join(REF_CSRC, rel), "rb").read()).hexdigest()
        if got != want:
            raise RuntimeError(f"{rel}: sha256 {got} is not the surveyed reference file's ({want}); refusing to compile it")
    from torch.utils import cpp_extension
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="drc_ref_")
    try:
        srcs = []
        for rel, patches in PATCHES.items():
            lines = open(os.path.join(REF_CSRC, rel)).read().split("\n")
            for ln, old, new in patches:
                if old not in lines[ln - 1]:
                    raise RuntimeError(f"{rel}:{ln} does not contain {old!r}: the reference differs from the surveyed one")
                lines[ln - 1] = lines[ln - 1].replace(old, new)
            dst = os.path.join(tmp, os.path.basename(rel))
            open(dst, "w").write("\n".join(lines))
            srcs.append(dst)
        srcs.append(os.path.join(HERE, "ref_binding.cpp"))
        bdir = os.path.join(tmp, "build")
        os.makedirs(bdir)
        cpp_extension.load(name=NAME, sources=srcs, extra_include_paths=[REF_CSRC], build_directory=bdir,
                           extra_cflags=["-O2", "-Wno-deprecated-declarations"], verbose=verbose, is_python_module=True)
        for so in glob.glob(os.path.join(bdir, NAME + "*.so")):
            shutil.copy(so, OUT)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return built_module_path()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
