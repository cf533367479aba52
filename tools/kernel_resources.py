"""Developer tool (no GPU needed): register / scratch / occupancy figures of every kernel of libidkpt.so, from the compiler's own
resource-usage remarks (hipcc -Rpass-analysis=kernel-resource-usage on the product's translation unit, same flags as the build).
usage: python tools/kernel_resources.py [filter-substring] [--dev] [--all]"""
import os
import re
import subprocess
import sys
import shutil
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from idkengine_amd import build as B  # noqa: E402


def resources(developer=False):
    tmp = tempfile.mkdtemp()
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-fPIC", "-shared", "-fvisibility=hidden")]
    cmd = [B._hipcc()] + flags + (["-DIDKPT_DEVELOPER"] if developer else []) + ["--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", os.path.join(B.CSRC, "idkpt.hip"), "-o", os.path.join(tmp, "dev.o")]
    txt = subprocess.run(cmd, cwd=B.CSRC, capture_output=True, text=True).stderr
    shutil.rmtree(tmp, ignore_errors=True)
    rows = []
    filt = shutil.which("c++filt")
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].split(" [")[0].strip()
        g = lambda k: int((re.search(k + r": (\d+)", b) or [None, "-1"])[1])  # noqa: E731
        if filt:
            name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() or name
        rows.append(dict(name=name, vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("SGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"), occ=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]")))
    return rows


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = resources(developer="--dev" in sys.argv)
    for r in rows:
        if "bvhgpu" in r["name"] and "--all" not in sys.argv:
            continue
        if args and not any(a in r["name"] for a in args):
            continue
        print(f"{r['name'][:120]:120s} VGPR {r['vgpr']:4d} SGPR {r['sgpr']:4d} scratch {r['scratch']:4d} waves/SIMD {r['occ']:2d}")
