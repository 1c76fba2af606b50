"""Registers, scratch and LDS of every kernel in a hipcc object / shared library (gfx950 code object metadata).
    python tools/kernel_resources.py <file.o|file.so> [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def resources(path):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path])
        data = open(fat, "rb").read()  # a linked library holds one bundle per translation unit, back to back
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
        notes = ""
        for i, st in enumerate(starts):
            part = os.path.join(d, "part%d.bin" % i)
            with open(part, "wb") as f:
                f.write(data[st:starts[i + 1] if i + 1 < len(starts) else len(data)])
            subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
            notes += subprocess.check_output([LLVM + "llvm-readelf", "--notes", co], text=True)
    out = []
    for block in notes.split("- .agpr_count:")[1:]:
        def field(name):
            m = re.search(r"\." + name + r":\s+(\S+)", block)
            return m.group(1) if m else "?"
        name = subprocess.check_output(["c++filt", field("name")], text=True).strip()
        out.append((name, int(field("vgpr_count")), int(block.split()[0]), int(field("sgpr_count")),
                    int(field("private_segment_fixed_size")), int(field("group_segment_fixed_size"))))
    return out


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, v, ag, sg, scratch, lds in sorted(resources(sys.argv[1])):
        if flt in name:
            print("%-90s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d" % (name[:90], v, ag, sg, scratch, lds))
