"""Builds libaether_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

The shared object is written next to the sources (aether_amd/csrc/libaether_hip.so) so that it travels
to the GPU box with the repository snapshot; nothing is JIT-compiled at import time.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libaether_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"]


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _code_only(text: str) -> str:
    """Source text without comments and with white space collapsed: what the compiler sees, give or take (string literals that
    contain comment markers would be clipped too — the same way every time, which is all a digest needs)."""
    import re
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"\s+", " ", text).strip()


def source_digest() -> str:
    """sha256 (first 16 hex digits) over the CODE of the kernel sources and the C header (comments and white space do not count:
    editing a comment must not make a measured profile look stale): stamps profiles so that a summary taken from an older build
    of the kernels is recognised as such (bench.py's roofline.traffic, tools/summarize_rocprof.py)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(list(_sources()) + list(CSRC.glob("*.hpp")) + [CSRC.parent.parent / "include" / "aether_hip.h"]):
        h.update(f.name.encode())
        h.update(_code_only(f.read_text()).encode())
    return h.hexdigest()[:16]


def _stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = list(_sources()) + list(CSRC.glob("*.hpp")) + [CSRC.parent.parent / "include" / "aether_hip.h"]
    return any(d.stat().st_mtime > t for d in deps)


def _compile(src: Path, obj: Path):
    cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")


def build_native(force: bool = False, verbose: bool = False) -> Path:
    """Compile every .hip translation unit for gfx950 and link libaether_hip.so. Returns its path."""
    if not force and not _stale():
        return LIB
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    srcs = _sources()
    objs = [objdir / (s.stem + ".o") for s in srcs]
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        list(ex.map(lambda so: _compile(*so), zip(srcs, objs)))
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} from {len(srcs)} sources", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build_native(force="--force" in sys.argv, verbose=True)
