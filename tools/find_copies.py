#!/usr/bin/env python3
"""Dev tool: which host lines issue copy / fill ops (device memcpy / memset launches) during Performer training steps.

    python tools/find_copies.py            # on a GPU box; prints op, count over the measured steps, and the innermost frames inside this repo

A TorchDispatchMode logs every aten copy / fill / clone with the Python stack; the hand-scheduled backward runs on autograd's device thread, where
thread-local modes are not active, so `_StackChain.backward` is wrapped to enter the mode there too."""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from synthanatomy_amd.networks.transformers import performer as P  # noqa: E402

SITES = collections.Counter()
KEYS = ("copy", "fill", "zero", "clone", "_to_copy", "cat", "index_put", "slice_scatter")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in KEYS):
            fr = [f"{os.path.relpath(f.filename, ROOT)}:{f.lineno}" for f in traceback.extract_stack(limit=30)
                  if ROOT in f.filename and "find_copies" not in f.filename][-3:]
            SITES[(name, " <- ".join(reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))


def main():
    args = argparse.Namespace(dtype="bf16", performer_shape="10,14,10", performer_batch=6, steps=2, warmup=0, sampling=False)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    orig = P._StackChain.backward

    def wrapped(self, dy, tape):
        with Log():
            return orig(self, dy, tape)

    P._StackChain.backward = wrapped
    with Log():
        bench.bench_performer(args, 0, 1, dev)
    for (name, where), c in SITES.most_common(45):
        print(f"{c:6d}  {name:34s} {where}")


if __name__ == "__main__":
    main()
