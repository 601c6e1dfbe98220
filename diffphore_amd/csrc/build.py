#!/usr/bin/env python3
"""Build libdiffphore_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output so it ships with gpurun."""
import glob
import hashlib
import re
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "libdiffphore_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wno-unused-result"]


_CODE = re.compile(r'"(?:\\.|[^"\\\n])*"|\'(?:\\.|[^\'\\\n])*\'|//[^\n]*|/\*.*?\*/', re.S)


def _code_only(text: str) -> str:
    """the source without comments, every run of white space as one blank (string and character literals kept as they are): what the hash below is
    taken over - a reworded comment or a re-wrapped line is not a different kernel"""
    text = _CODE.sub(lambda m: m.group(0) if m.group(0)[0] in "\"'" else " ", text)
    return " ".join(text.split())


def source_hash() -> str:
    """sha256 over the CODE of the kernel sources (csrc/*.hip, csrc/*.h, include/*.h; comments and white space do not count), first 16 hex
    digits: compiled into the library (dp_build_info) and used to key committed PMC profiles to the sources they were measured on."""
    h = hashlib.sha256()
    root = os.path.normpath(os.path.join(HERE, "..", ".."))
    for f in sorted(glob.glob(os.path.join(HERE, "*.hip")) + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(root, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(_code_only(open(f, "r", encoding="utf-8").read()).encode())
    return h.hexdigest()[:16]


def _stale(objs_src):
    if not os.path.exists(OUT):
        return True
    stamp = os.path.join(HERE, "dp_host.o.hash")
    if not os.path.exists(stamp) or open(stamp).read().strip() != source_hash():
        return True
    t = os.path.getmtime(OUT)
    deps = objs_src + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "..", "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(src, seen=None):
    """`src` plus the quoted headers it includes, transitively (a kernel file is rebuilt only when one of ITS headers changed: the
    node-side conv takes ~2 minutes to compile, the edge-side one ~4)."""
    import re
    seen = seen if seen is not None else set()
    if src in seen or not os.path.exists(src):
        return seen
    seen.add(src)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(src).read(), flags=re.M):
        _deps(os.path.normpath(os.path.join(os.path.dirname(src), inc)), seen)
    return seen


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    if not srcs:
        raise RuntimeError("no .hip sources found")
    if not force and not _stale(srcs):
        return OUT
    objs = []
    procs = []
    shash = source_hash()
    stamp = os.path.join(HERE, "dp_host.o.hash")          # dp_host.o carries the source hash (dp_build_info): rebuilt whenever it changes
    host_stale = not os.path.exists(stamp) or open(stamp).read().strip() != shash
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        is_host = os.path.basename(s) == "dp_host.hip"
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(d) for d in _deps(s)) or (is_host and host_stale):
            # the two-limb fp16 kernels spell v_fma_mixlo_f16 / v_fma_mixhi_f16 as scalar C (dp_conv_node_f16_impl.h): SLP would re-pack them
            extra = ["-fno-slp-vectorize"] if os.path.basename(s).startswith("dp_conv_node_f16_k") else []
            cmd = [HIPCC] + FLAGS + extra + ([f'-DDP_SOURCE_HASH="{shash}"'] if is_host else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    # exports: the dp_* entry points only (exports.map) - the C++ launchers that connect the translation units are not part of the ABI
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(HERE, "exports.map")] + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(stamp, "w").write(shash + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
