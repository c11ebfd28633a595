"""Build librrt_hip.so in-tree: hipcc --offload-arch=gfx950 on csrc/*.hip (no GPU needed).

    python rrt-mil_amd/build.py [--force] [--verbose]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librrt_hip.so")
STAMP = os.path.join(HERE, "csrc", ".build_stamp")
SOURCES = ["ln_partition.hip", "cast16.hip", "linear_f32.hip", "region_attn.hip", "rmsa_fused.hip", "rmsa_fused16.hip", "rmsa_pair16.hip", "rmsa_fused_x3.hip", "crmsa.hip", "epeg_variants.hip", "mil_pool.hip", "linear_bwd.hip", "ln_bwd.hip", "attn_bwd.hip", "crmsa_bwd.hip", "peg.hip",
           "api.hip"]
HEADERS = ["common.h", "internal.h", "fused16.h", os.path.join("..", "..", "include", "rrt_hip.h")]
# -amdgpu-mfma-vgpr-form: gfx950 has one unified VGPR/AGPR file; keep MFMA accumulators in VGPRs so the
# softmax / rescale VALU code does not shuttle them through v_accvgpr_read/write (hazard stalls).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-Wall", "-Wno-unused-function"]


def _hdr_digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h


def _src_digest(src):
    """digest of one translation unit: flags + every header + the source (headers are few: no dependency scan)"""
    h = _hdr_digest()
    with open(os.path.join(CSRC, src), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def _digest():
    h = hashlib.sha256()
    for f in SOURCES:
        h.update(_src_digest(f).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile what changed (force: everything); returns the library path.  Each object carries the digest of its
    source + headers + flags in a side file (csrc/<name>.o.stamp), the library the digest of all of them."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dig:
                return LIB
    if not os.path.exists(hipcc):
        raise RuntimeError(f"hipcc not found at {hipcc}; cannot build librrt_hip.so")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        sd = _src_digest(src)
        if not force and os.path.exists(obj) and os.path.exists(obj + ".stamp"):
            with open(obj + ".stamp") as fh:
                if fh.read().strip() == sd:
                    continue
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        if os.path.exists(obj + ".stamp"):
            os.remove(obj + ".stamp")
        procs.append((src, obj, sd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for src, obj, sd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(f"hipcc failed on {src}:\n{out.decode()}")
            continue
        with open(obj + ".stamp", "w") as fh:
            fh.write(sd)
        if verbose and out:
            print(out.decode())
    if failed:
        raise RuntimeError("\n".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
