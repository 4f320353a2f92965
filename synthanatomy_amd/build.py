"""Builds libsynthanatomy_hip.so (the C-ABI of include/synthanatomy_hip.h) for gfx950 with hipcc, in-tree.

    python -m synthanatomy_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# SA_BUILD_VARIANT=name (dev): a second library next to the product one -- libsynthanatomy_hip_<name>.so from _obj_<name>/ with SA_EXTRA_HIPCC_FLAGS --,
# loaded with SA_HIP_LIB=<path> for A/B runs (e.g. -DSA_PP_DEBUG_VARIANTS, -DSA_TIMING); the product build is untouched
_VARIANT = os.environ.get("SA_BUILD_VARIANT", "")
LIB = os.path.join(HERE, f"libsynthanatomy_hip{'_' + _VARIANT if _VARIANT else ''}.so")
OBJ = os.path.join(HERE, f"_obj{'_' + _VARIANT if _VARIANT else ''}")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off", "-Wno-unused-value"]
FLAGS += os.environ.get("SA_EXTRA_HIPCC_FLAGS", "").split()  # dev: e.g. -DSA_PP_DEBUG_VARIANTS


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path):
    h = hashlib.sha1()
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    if not path.endswith(("conv_fprop.hip", "conv_fprop_f16.hip")):
        hdrs = [h for h in hdrs if not h.endswith(("conv_fprop_common.h", "conv_fprop_kernels.h"))]     # only the forward translation units depend on them
    if not path.endswith(("local_attn.hip", "favor_fused.hip")):
        hdrs = [h for h in hdrs if not h.endswith("local_attn_split.h")]
    for dep in [path, *hdrs, os.path.join(HERE, "..", "include", "synthanatomy_hip.h")]:
        with open(dep, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + EXTRA.get(os.path.basename(path), [])).encode())
    return h.hexdigest()


# per-file flags (none since dense.hip and its compiler-internal -mllvm flag left the tree in round 6)
EXTRA = {}


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".sha1"
    dg = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg:
        return obj, False
    cmd = [HIPCC, *FLAGS, *EXTRA.get(src, []), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dg)
    return obj, True


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(_compile, _sources()))
    objs = [o for o, _ in res]
    for f in os.listdir(OBJ):                     # objects of sources that left the tree
        if f.endswith(".o") and os.path.join(OBJ, f) not in objs:
            os.remove(os.path.join(OBJ, f))
            res.append((None, True))
    if any(c for _, c in res) or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[synthanatomy_amd.build] linked {LIB} ({len(objs)} objects, {sum(c for _, c in res)} recompiled)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
