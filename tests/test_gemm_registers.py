"""The 256 x 256 SGEMM tile has 128 registers per lane, 64 of them accumulators, and keeps K-tiles in flight in registers that
inline-asm buffer loads write behind the compiler's back (csrc/gemm_tile_body.inc).  If the allocator spills one of those ring
registers it saves it BEFORE its load has landed and reuses the physical register - the late load then lands on another value
(round 3's "memory access fault" with three tiles in flight; csrc/gemm.hip: launch_layout).  So the shipped instantiations must
use no scratch at all.  hipcc cross-compiles without a GPU: only these four kernels are compiled here (a few seconds)."""
import os
import re
import shutil
import subprocess
import pytest
from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
CSRC = os.path.join(ROOT, "lightgrad_amd", "csrc")
BIG = ["sgemm_mfma<256, 256, 32, 4, 4, %s, %s, true, true, 2, 1, 0>" % (a, b) for a in ("true", "false") for b in ("true", "false")]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.timeout(600)
def test_the_large_tile_uses_no_scratch(tmp_path):
    text = open(os.path.join(CSRC, "gemm.hip")).read()
    cut = text.index("// Two independent products in ONE launch")
    shutil.copytree(CSRC, tmp_path / "pkg" / "csrc", ignore=shutil.ignore_patterns("*.o", "*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")     # the sources include ../../include/lghip.h
    unit = tmp_path / "pkg" / "csrc" / "big_only.hip"
    unit.write_text(text[:cut] + "".join("template __global__ void %s(GemmArgs);\n" % k for k in BIG) + "}  // namespace lg\n")
    out = tmp_path / "big.s"
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-o", str(out), "big_only.hip"], cwd=str(tmp_path / "pkg" / "csrc"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    asm = out.read_text()
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S)
    assert len(kernels) == 4, [k for k, _ in kernels]
    for name, body in kernels:
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        assert scratch == 0, "%s: %d bytes of scratch per lane" % (name, scratch)
        assert vgprs <= 128, "%s: %d registers: 16 waves of it do not fit a CU" % (name, vgprs)
    assert "scratch_load" not in asm and "scratch_store" not in asm
