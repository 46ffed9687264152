"""Register / scratch / LDS budgets of the hot kernels, from the compiler's own report (hipcc cross-compiles gfx950
without a GPU).  What the designs in DESIGN.md 4 rest on: no kernel of the path spills to scratch memory; the sort keeps
three 256-thread workgroups per CU (<= 168 VGPRs, <= 53 KB LDS), the forward four waves per SIMD (<= 128 VGPRs), the
backward's 12-wave workgroup fits one CU (<= 170 VGPRs incl. AGPRs, <= 160 KB LDS)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "humangaussian_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

LIMITS = {  # kernel: (max VGPRs incl. AGPRs, max LDS bytes per workgroup)
    "hgs_k_sort_lds": (168, 54 * 1024),
    "hgs_k_render_fwd_store": (128, 16 * 1024),
    "hgs_k_render_fwd_nostore": (128, 16 * 1024),
    "hgs_k_render_bwd": (170, 160 * 1024),
    "hgs_k_pair_reduce_em": (128, 24 * 1024),
    "hgs_k_sort_lds_ch": (168, 54 * 1024),
    "hgs_k_pair_reduce_ch": (128, 24 * 1024),
    "hgs_k_preprocess_fwd": (128, 64 * 1024),
    "hgs_k_fill": (128, 64 * 1024),
    "hgs_k_tiles": (128, 16 * 1024),
}


def report(src):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, "o.o")]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-3000:]
    out, cur = {}, None
    for line in p.stdout.splitlines():
        m = re.search(r"remark: +Function Name: (\w+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_scratch_and_register_budgets_of_the_hot_kernels():
    with ThreadPoolExecutor(2) as ex:
        reps = list(ex.map(report, ["api.hip", "render_bwd.hip"]))
    rep = {}
    for r in reps:
        rep.update(r)
    kernels = {k: v for k, v in rep.items() if k.startswith("hgs_k_")}
    assert len(kernels) >= 30, sorted(kernels)
    # no kernel of the path touches scratch memory (the many-view loop form of the SH-3 preprocess backward parks
    # registers in AGPRs - "VGPRs Spill" without scratch - and runs one wave per SIMD: the one slow corner)
    spilled = {k: v for k, v in kernels.items() if v.get("ScratchSize", 0)}
    assert not spilled, spilled
    to_agprs = sorted(k for k, v in kernels.items() if v.get("VGPRs Spill", 0))
    assert to_agprs in ([], ["hgs_k_preprocess_bwd_d3"]), to_agprs
    for name, (vmax, ldsmax) in LIMITS.items():
        k = kernels[name]
        regs = max(k.get("VGPRs", 0), 0) + k.get("AGPRs", 0)
        assert regs <= vmax, (name, k)
        assert k.get("LDS Size", 0) <= ldsmax, (name, k)
    assert kernels["hgs_k_sort_lds"].get("Occupancy", 0) >= 3 and kernels["hgs_k_sort_lds_ch"].get("Occupancy", 0) >= 3 and kernels["hgs_k_render_fwd_store"].get("Occupancy", 0) >= 4
