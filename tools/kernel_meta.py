#!/usr/bin/env python3
"""Per-kernel resource table of libdfe_hip.so without a GPU: the code objects are carved out of the library's .hip_fatbin
section and their AMDGPU metadata notes (msgpack) are read -- registers, spills, static LDS, scratch.  Round 6 lost 3 ms of
step time to four bytes of static LDS in k_wino_fwd16 (two blocks per CU -> one); tests/test_api_cpu.py holds the
occupancy-critical kernels to what this prints.

    python tools/kernel_meta.py [pattern]            (default: every kernel, sorted by VGPRs)
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

import msgpack

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unsupervised_depth_opticalflow_egomotion_amd", "libdfe_hip.so")
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"


def _notes(elf):
    """AMDGPU metadata (NT_AMDGPU_METADATA = 32) of one ELF64 code object."""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        off = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, off + 4)
        if sh_type != 7:        # SHT_NOTE
            continue
        s_off, s_size = struct.unpack_from("<QQ", elf, off + 0x18)
        p = s_off
        while p + 12 <= s_off + s_size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):
                return msgpack.unpackb(desc, raw=False, strict_map_key=False)
    return None


def kernels(lib=LIB):
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, lds, scratch, max_wg}] of every gfx950 kernel in the library."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([OBJCOPY, "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
    out = []
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)] + [len(data)]
    for a, b in zip(starts[:-1], starts[1:]):
        e = data.find(b"\x7fELF", a, b)
        if e < 0:
            continue
        md = _notes(data[e:b])
        if not md:
            continue
        for k in md.get("amdhsa.kernels", []):
            out.append(dict(name=k[".name"], vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0), sgpr=k.get(".sgpr_count", 0),
                            vgpr_spill=k.get(".vgpr_spill_count", 0), sgpr_spill=k.get(".sgpr_spill_count", 0),
                            lds=k.get(".group_segment_fixed_size", 0), scratch=k.get(".private_segment_fixed_size", 0),
                            max_wg=k.get(".max_flat_workgroup_size", 0)))
    return out


def demangle(name):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        return name


if __name__ == "__main__":
    pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    rows = [k for k in kernels() if pat is None or pat.search(k["name"])]
    print("%-70s %5s %5s %5s %6s %7s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "spill", "lds B", "scratch"))
    for k in sorted(rows, key=lambda k: -k["vgpr"]):
        print("%-70s %5d %5d %5d %6d %7d %6d" % (demangle(k["name"]).replace("void ", "").replace("dfe::", "")[:70], k["vgpr"], k["agpr"],
                                               k["sgpr"], k["vgpr_spill"] + k["sgpr_spill"], k["lds"], k["scratch"]))
