#!/usr/bin/env python3
"""Static resources of every gfx950 kernel in csrc/ (dev tool; runs without a GPU).

    python tools/kernel_resources.py [file.hip ...]          # default: every csrc/*.hip (conv_fprop.hip takes ~5 minutes)

Compiles each source device-only (`hipcc --cuda-device-only --no-gpu-bundle-output`) and reads the kernel descriptors' metadata with llvm-readelf: static LDS
bytes, VGPRs (incl. AGPRs), spills, work-group size, and from those the resident work-groups per CU the STATIC resources allow (160 KiB LDS, 512 VGPRs per
SIMD lane, 4 SIMDs; dynamic LDS comes on top at launch).  Written after round 3 found `favor_prepass_kernel` at one block per CU instead of two: 32 bytes of
static LDS beside exactly 80 KiB of dynamic LDS."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "synthanatomy_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=off", "-Wno-unused-value", "--cuda-device-only", "--no-gpu-bundle-output"]


def kernels(src):
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "k.co")
        subprocess.run([HIPCC, *FLAGS, "-c", src, "-o", co], check=True, capture_output=True)
        notes = subprocess.run([READELF, "--notes", co], check=True, capture_output=True, text=True).stdout
    names = subprocess.run(["c++filt"], input="\n".join(re.findall(r"^\s+\.name:\s+(\S+)", notes, re.M)), capture_output=True, text=True).stdout.split("\n")
    out = []
    for blk, name in zip(re.split(r"^\s+- \.agpr_count:", notes, flags=re.M)[1:], names):
        f = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))
        out.append((re.sub(r"\(.*$", "", name).replace("sa::", "").replace("void ", ""), f("group_segment_fixed_size"), f("vgpr_count"), f("vgpr_spill_count"),
                    f("max_flat_workgroup_size")))
    return out


def main():
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    print(f"{'kernel':78s} {'LDS B':>7s} {'VGPR':>5s} {'spill':>5s} {'wg':>5s} {'wg/CU (static)':>14s}")
    for src in files:
        print(f"== {os.path.basename(src)}")
        for name, lds, vgpr, spill, wg in kernels(src):
            waves = (wg + 63) // 64
            per_simd = max(1, 512 // max(vgpr, 1))                       # waves per SIMD the register file admits (<= 8)
            by_vgpr = min(per_simd, 8) * 4 // waves if waves <= 4 * min(per_simd, 8) else 0
            by_lds = (160 * 1024) // lds if lds else 99
            print(f"{name[:78]:78s} {lds:7d} {vgpr:5d} {spill:5d} {wg:5d} {min(by_vgpr, by_lds, 32):14d}")


if __name__ == "__main__":
    main()
