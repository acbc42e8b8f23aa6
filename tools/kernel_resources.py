"""Register / scratch / LDS budget of every kernel in a built library, from the code objects embedded in it.

    python tools/kernel_resources.py [wiki-grx-gym_amd/csrc/libgrx_hip.so] [--json profiles/rNN_kernel_resources.json]

Walks the __CLANG_OFFLOAD_BUNDLE__ containers of the .so, takes every gfx950 code object and reads the AMDGPU metadata notes
(llvm-readelf --notes): vgpr / agpr counts, spilled VGPRs, private segment (scratch) bytes per lane, LDS bytes.  The numbers
VERDICT r3 asked to drive to zero -- `vgpr_spill_count`, `private_segment_fixed_size` of the step kernels -- come from here."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
DEMANGLE = "c++filt"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def kernels_of(co_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(co_bytes)
        path = f.name
    try:
        txt = subprocess.run([READELF, "--notes", path], capture_output=True, text=True, check=True).stdout
    finally:
        os.unlink(path)
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda key: (re.search(r"\." + key + r":\s+(\S+)", blk) or [None, None])[1]
        sym = g("name")
        name = subprocess.run([DEMANGLE, sym], capture_output=True, text=True).stdout.strip() if sym else "?"
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
        out.append({"kernel": name, "vgpr": int(g("vgpr_count")), "agpr": int(g("agpr_count")), "sgpr": int(g("sgpr_count")),
                    "vgpr_spill": int(g("vgpr_spill_count")), "sgpr_spill": int(g("sgpr_spill_count")),
                    "scratch_bytes": int(g("private_segment_fixed_size")), "lds_bytes": int(g("group_segment_fixed_size")),
                    "max_flat_workgroup_size": int(g("max_flat_workgroup_size"))})
    return out


def main():
    argv = list(sys.argv[1:])
    out_json = None
    if "--json" in argv:
        i = argv.index("--json"); out_json = argv[i + 1]; del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wiki-grx-gym_amd", "csrc", "libgrx_hip.so")
    rows = []
    for co in code_objects(open(lib, "rb").read()):
        rows += kernels_of(co)
    rows.sort(key=lambda r: r["kernel"])
    print(f"{'kernel':58s} vgpr agpr spill scratch    lds")
    for r in rows:
        print(f"{r['kernel'][:58]:58s} {r['vgpr']:4d} {r['agpr']:4d} {r['vgpr_spill']:5d} {r['scratch_bytes']:7d} {r['lds_bytes']:6d}")
    if out_json:
        with open(out_json, "w") as f:
            json.dump({"library": os.path.basename(lib), "kernels": rows}, f, indent=1)


if __name__ == "__main__":
    main()
