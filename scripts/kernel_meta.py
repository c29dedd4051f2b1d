#!/usr/bin/env python3
"""Per-kernel resource metadata of the gfx950 code object inside libtransgo_hip.so (no GPU needed): VGPR / AGPR / SGPR counts, LDS,
private-segment (scratch) size, spill counts, dynamic stack -- the `amdhsa.kernels` note of the code object.
    python3 scripts/kernel_meta.py [substring ...]      (or --lib PATH)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib):
    tmp = tempfile.mkdtemp(prefix="tgco_")
    out = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", f"--input={lib}"], capture_output=True, text=True)
    objs = []
    if out.returncode == 0:
        for t in out.stdout.split():
            if "amdgcn" in t:
                o = os.path.join(tmp, t.replace("/", "_") + ".co")
                subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={lib}", f"--targets={t}", f"--output={o}"], check=True)
                objs.append(o)
    if not objs:                                   # a shared library: the fat binary sits in .hip_fatbin
        fb = os.path.join(tmp, "fatbin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True)
        data = open(fb, "rb").read()
        # clang offload bundles: magic, then entries (offset, size, triple)
        import struct
        pos = 0
        while True:
            pos = data.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
            if pos < 0:
                break
            n = struct.unpack_from("<Q", data, pos + 24)[0]
            q = pos + 32
            for _ in range(n):
                off, size, tl = struct.unpack_from("<QQQ", data, q)
                triple = data[q + 24:q + 24 + tl].decode()
                q += 24 + tl
                if "amdgcn" in triple and size:
                    o = os.path.join(tmp, f"co_{pos}_{len(objs)}.co")
                    open(o, "wb").write(data[pos + off:pos + off + size])
                    objs.append(o)
            pos += 24
    return objs


def kernels(lib):
    res = []
    for o in code_objects(lib):
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", o], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.", txt):
            m = re.search(r"\.name:\s+(\S+)", blk)
            if not m or ".vgpr_count" not in blk:
                continue
            g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            res.append(dict(name=name, vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"),
                            scratch=g("private_segment_fixed_size"), vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"),
                            dyn_stack=g("uses_dynamic_stack")))
    return res


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = os.path.join(ROOT, "transgo_amd", "libtransgo_hip.so")
    if args and args[0] == "--lib":
        lib = args[1]; args = args[2:]
    for k in kernels(lib):
        if not args or any(a in k["name"] for a in args):
            nm = re.sub(r"^void \(anonymous namespace\)::|^void tg::", "", k["name"]).split("(")[0]
            print(f"{nm:70s} vgpr {k['vgpr']:>4} agpr {k['agpr']:>4} sgpr {k['sgpr']:>4} lds {k['lds']:>7} scratch {k['scratch']:>6} "
                  f"vspill {k['vspill']:>4} sspill {k['sspill']:>4} dyn_stack {k['dyn_stack']}")
