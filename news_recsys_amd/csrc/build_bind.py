#!/usr/bin/env python3
"""Builds news_recsys_amd/lib/nrx_bind*.so (csrc/nrx_bind.cpp: the compiled host binding of the module path) in-tree with g++
against the installed libtorch.  Called by __graft_entry__.build() and `make -C news_recsys_amd/csrc bind`."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def target() -> str:
    return os.path.join(ROOT, "news_recsys_amd", "lib", "nrx_bind" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build(force: bool = False) -> str:
    import torch
    from torch.utils import cpp_extension as ce
    src, out = os.path.join(HERE, "nrx_bind.cpp"), target()
    hdr = os.path.join(ROOT, "include", "nrx_embed.h")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    libdir = ce.library_paths()[0]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", src, "-o", out,
           "-DTORCH_EXTENSION_NAME=nrx_bind", "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-I" + os.path.join(ROOT, "include"), "-I" + sysconfig.get_paths()["include"]]
    cmd += ["-isystem" + p for p in ce.include_paths()]
    cmd += ["-L" + libdir, "-Wl,-rpath," + libdir, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python"]
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
