"""Compare the gfx950 assembly of every kernel of two builds (refactors that must not change the device code).

  python tools/isa_diff.py dump OUTDIR [-DFLAG ...]     # hipcc -S --cuda-device-only of both translation units
  python tools/isa_diff.py cmp DIR_A DIR_B              # per kernel: equal / differs (instruction count of each side)

Kernel bodies are compared from their entry label to `s_endpgm`-terminated `.Lfunc_end`, with local label numbers
normalised (they shift when an unrelated function is added or removed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "humangaussian_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S", "--cuda-device-only"]


def dump(out, extra):
    os.makedirs(out, exist_ok=True)
    procs = [subprocess.Popen(["hipcc"] + FLAGS + extra + ["-o", os.path.join(out, s), os.path.join(CSRC, u)])
             for u, s in (("api.hip", "api.s"), ("render_bwd.hip", "bwd.s"))]
    assert all(p.wait() == 0 for p in procs)


def kernels(path):
    res, name, body = {}, None, []
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L") and name is None and ("hgs_k_" in m.group(1)):
            name, body = m.group(1), []
            continue
        if name is not None:
            if line.startswith(".Lfunc_end"):
                res[name] = body
                name = None
                continue
            s = line.split(";")[0].strip()
            if not s or s.startswith("."):
                if s.startswith(".LBB"):
                    body.append("LABEL")
                continue
            body.append(re.sub(r"\.LBB\d+_\d+", "LBB", s))
    return res


def cmp(a, b):
    same = True
    for f in ("api.s", "bwd.s"):
        ka, kb = kernels(os.path.join(a, f)), kernels(os.path.join(b, f))
        for k in sorted(set(ka) | set(kb)):
            if k not in ka or k not in kb:
                print(f"{k}: only in {'B' if k not in ka else 'A'}")
                same = False
            elif ka[k] == kb[k]:
                print(f"{k}: equal ({len(ka[k])} lines)")
            else:
                print(f"{k}: DIFFERS ({len(ka[k])} vs {len(kb[k])} lines)")
                same = False
    return same


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3:])
    else:
        sys.exit(0 if cmp(sys.argv[2], sys.argv[3]) else 1)
