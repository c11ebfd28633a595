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
# -amdgpu-kernarg-preload-count: the leading kernel arguments (pointers, sizes: up to 16 dwords) arrive in SGPRs with the
#   wave instead of through a cold scalar load at its entry -- 0.1-0.2 us on each of the latency-bound launches, one bag in
#   flight 78.4 -> 77.6 us in bf16 (round 5, tools/experiments/ab_kernarg_preload.sh); the compiler keeps the loading
#   prologue for firmware that does not preload.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-Wall", "-Wno-unused-function"]


# Packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) are OFF for every translation unit unless the file is
# listed in PACKED_FP32_OK below -- and there only inside the kernels named for it.
# Round 5 found what the round-2 "lanes 48..63" mis-sums were: compiler-formed packed fp32 operations (seen: the low lane of a
# v_pk_fma_f32 whose op_sel takes the HIGH dword of a source pair) return run-to-run different values when the wave shares
# its SIMD with bf16-MFMA waves of ANOTHER bag's kernel -- crmsa_combine_parts_kernel's representative n = 1, components x / z,
# bf16 mode only, two bags in flight only; never alone, never next to fp32-MFMA kernels (tools/experiments/dbg_abi2.py;
# tools/ubench/pk_fma_beside_bf16_mfma.hip).  With the feature off for the translation unit the same
# forwards are bit-identical.  Round 6 turned the rule around (default off, opt in per file AND per kernel) and made the build
# check it: after a unit is compiled its gfx950 code object is disassembled (llvm-objdump) and the build FAILS if a packed fp32
# instruction sits in a kernel that is not on the unit's list (today: anywhere) -- a new or edited kernel, or a compiler upgrade that ignores the
# feature flag, cannot bring the mis-sum back silently.  (The flag reaches the host compilation too, which says "not a
# recognized feature for this target" and ignores it: filtered from the build output.)
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# file -> kernels (substring of the demangled name) that may contain packed fp32.  EMPTY since round 6: with the feature off in
# every translation unit the headline is unchanged within run-to-run noise (fp32 5303 / 5314 against 5319 / 5318 slides/s, bf16
# 20.20 / 20.50 k against 20.01 / 20.35 k, same box, alternating: profiles/r06_nopk_ab.txt) -- next to MFMAs a v_pk_fma_f32 costs
# more than the two v_fma_f32 it replaces (MI355X_MICROARCH.md, "price of one filler beside MFMAs").  A kernel that wants them
# back gets an entry here AND __attribute__((target("packed-fp32-ops"))) on the kernel (the function-level attribute overrides
# the unit's flag in both directions: tested), and a soak run with four bf16 bags in flight.
PACKED_FP32_OK = {}
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
_PK_RE = __import__("re").compile(r"\bv_pk_(fma|mul|add)_f32\b")
_SYM_RE = __import__("re").compile(r"^[0-9a-f]+ <(.*)>:$")


def flags_for(src):
    return FLAGS + ([] if src in PACKED_FP32_OK else NO_PACKED_FP32)


def packed_fp32_census(obj, seen=None):
    """{demangled kernel name: number of v_pk_{fma,mul,add}_f32} over the gfx950 code objects bundled in a hipcc object file
    or in the linked library (one bundle per translation unit).  `seen` (a list) receives every function symbol met."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="rrt_pk_")
    outs = []
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:                          # (a host-only unit -- api.hip: launches only -- has none)
                outs.append(subprocess.run([OBJDUMP, "-d", "-C", os.path.join(tmp, f)], check=True, capture_output=True,
                                           text=True).stdout)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    census, name = {}, None
    for out in outs:
        for line in out.splitlines():
            m = _SYM_RE.match(line)
            if m:
                name = m.group(1)
                if seen is not None:
                    seen.append(name)
            elif name is not None and _PK_RE.search(line):
                census[name] = census.get(name, 0) + 1
    return census


def check_packed_fp32(src, obj):
    """Fail the build if a packed fp32 instruction sits outside the kernels PACKED_FP32_OK names for this file."""
    ok = PACKED_FP32_OK.get(src, ())
    bad = {k: n for k, n in packed_fp32_census(obj).items() if not any(pat in k for pat in ok)}
    if bad:
        lines = "\n".join(f"    {n:5d}  {k[:160]}" for k, n in sorted(bad.items(), key=lambda kv: -kv[1]))
        raise RuntimeError(f"{src}: packed fp32 instructions (v_pk_fma/mul/add_f32) in kernels that are not allowed to contain "
                           f"them (rrt-mil_amd/build.py::PACKED_FP32_OK; DESIGN.md section 9):\n{lines}")


def _hdr_digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h


def _src_digest(src):
    """digest of one translation unit: flags + every header + the source (headers are few: no dependency scan)"""
    h = _hdr_digest()
    h.update((" ".join(flags_for(src)) + repr(PACKED_FP32_OK.get(src))).encode())
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
        cmd = [hipcc, *flags_for(src), "-c", os.path.join(CSRC, src), "-o", obj]
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
        try:
            check_packed_fp32(src, obj)          # (the stamp is written only for an object that passed)
        except (RuntimeError, subprocess.CalledProcessError) as e:
            failed.append(str(e))
            continue
        with open(obj + ".stamp", "w") as fh:
            fh.write(sd)
        if verbose and out:
            print("\n".join(l for l in out.decode().splitlines() if "not a recognized feature for this target" not in l))
    if failed:
        raise RuntimeError("\n".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
