"""Register / spill / scratch numbers of the kernels in a hipcc object or shared library (no GPU needed).

    python tools/kernel_regs.py image-super-resolution_amd/csrc/ffsr_tok.o [name-filter-regex]

Finds the clang offload bundle inside the file, takes the gfx950 code object and prints the AMDGPU metadata notes."""
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + 24)
        off = pos + 32
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24:off + 24 + tl].decode()
            off += 24 + tl
            if "gfx950" in triple and size:
                yield blob[pos + o:pos + o + size]
        pos += 24


def main():
    blob = open(sys.argv[1], "rb").read()
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            if pat and not pat.search(name):
                continue
            print(f"vgpr {g('vgpr_count'):>3} spill {g('vgpr_spill_count'):>3} sgpr-spill {g('sgpr_spill_count'):>3} "
                  f"scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}  {name[:150]}")


if __name__ == "__main__":
    main()
