"""Register / LDS / scratch usage of the device kernels in libazsp.so (reads the gfx950 code object out of the fat binary with
llvm-readelf --notes).  usage: python tools/kernel_resources.py [substring of the kernel name]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_object(lib):
    data = open(lib, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        trip = data[off:off + tl].decode()
        off += tl
        if "gfx950" in trip:
            return data[i + o:i + o + sz]
    raise SystemExit("no gfx950 code object in " + lib)


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object(os.path.join(ROOT, "alpha_zero_amd", "libazsp.so")))
        f.flush()
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True, check=True).stdout
        names = subprocess.run(["c++filt"], input="\n".join(re.findall(r"\.name:\s+(\S+)", notes)), capture_output=True,
                               text=True).stdout.split("\n")
    k = 0
    for blk in notes.split("- .agpr_count:")[1:]:
        name = names[k] if k < len(names) else "?"
        k += 1
        if pat not in name:
            continue
        g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
        agpr = int(blk.split()[0])
        print(f"{name[:110]:110s} vgpr+agpr {g('vgpr_count'):3d} (agpr {agpr:3d}) sgpr {g('sgpr_count'):3d} "
              f"lds {g('group_segment_fixed_size'):6d} scratch {g('private_segment_fixed_size')}")


if __name__ == "__main__":
    main()
