#!/bin/bash
# tools/build_ablation.sh <name> <extra hipcc flags...>  -> tools/_abl/librrt_<name>.so (git-ignored; ships with gpurun)
# Incremental: objects are kept under tools/_abl/obj_<name>/ with a stamp of (source + headers + flags); only what changed
# is recompiled.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p tools/_abl/obj_$name
python - "$name" "$@" <<'PY'
import hashlib, os, subprocess, sys
sys.path.insert(0, "rrt-mil_amd")
import build
name, extra = sys.argv[1], sys.argv[2:]
flags = [f for f in build.FLAGS if not f.startswith("-W")] + extra      # the product's flags (ABL_NO= drops one: e.g. the kernarg preload)
if os.environ.get("ABL_NO"):
    keep = []
    for f in flags:
        if os.environ["ABL_NO"] in f:
            if keep and keep[-1] == "-mllvm":
                keep.pop()
            continue
        keep.append(f)
    flags = keep
def file_flags(src):
    # the product's per-file rule (packed fp32 off unless the file is in build.PACKED_FP32_OK); ABL_PACKED=a.hip,b.hip turns
    # it ON for those files, ABL_NOPACKED=a.hip,b.hip OFF (A/B of one translation unit)
    on = src in build.PACKED_FP32_OK
    if src in os.environ.get("ABL_PACKED", "").split(","):
        on = True
    if src in os.environ.get("ABL_NOPACKED", "").split(","):
        on = False
    return [] if on else build.NO_PACKED_FP32
od = f"tools/_abl/obj_{name}"
hd = hashlib.sha256(" ".join(flags).encode())
for h in build.HEADERS:
    hd.update(open(os.path.join(build.CSRC, h), "rb").read())
procs, objs = [], []
for src in build.SOURCES:
    obj = os.path.join(od, src.replace(".hip", ".o"))
    objs.append(obj)
    d = hd.copy()
    d.update(" ".join(file_flags(src)).encode())
    d.update(open(os.path.join(build.CSRC, src), "rb").read())
    dig = d.hexdigest()
    st = obj + ".stamp"
    if os.path.exists(obj) and os.path.exists(st) and open(st).read().strip() == dig:
        continue
    if os.path.exists(st):
        os.remove(st)
    ff = file_flags(src)
    procs.append((src, st, dig, subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, *ff, "-c", os.path.join(build.CSRC, src), "-o", obj],
                                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
bad = False
for src, st, dig, p in procs:
    out, _ = p.communicate()
    if p.returncode:
        print(f"hipcc failed on {src}:\n" + "\n".join(l for l in out.decode().splitlines() if "not a recognized feature" not in l))
        bad = True
    else:
        open(st, "w").write(dig)
if bad:
    sys.exit(1)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", f"tools/_abl/librrt_{name}.so"], check=True)
print(f"tools/_abl/librrt_{name}.so ({len(procs)} recompiled)")
PY
