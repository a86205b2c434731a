"""Build-time audit of the hand-scheduled edge kernel (no GPU needed: hipcc cross-compiles gfx950).

The kernel issues its gathers and weight DMA from asm statements that hipcc does not count.
That is only safe while (a) no register that an in-flight asm load will write is read, moved or spilled by
compiler-generated code before the counted wait that names it, and (b) the kernel has no scratch (a spill of such
a register would silently save garbage).  Both are checked on the generated ISA."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "graph_weather_amd", "csrc", "gw_edge.hip")


@pytest.mark.timeout(600)
def test_edge_kernel_isa_has_no_scratch_and_no_hidden_load_hazards(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-c", SRC, "-o", "e.o", "-save-temps"]
    subprocess.run(cmd, cwd=tmp_path, check=True, capture_output=True)
    asm = tmp_path / "gw_edge-hip-amdgcn-amd-amdhsa-gfx950.s"
    text = asm.read_text()
    kernels = re.findall(r"^(_Z\w*edge_kernel\w*):", text, re.M)
    assert len(kernels) == 5, kernels
    for k in kernels:
        meta = text[text.index(".amdhsa_kernel " + k):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1))
        assert scratch == 0, f"{k}: {scratch} bytes of scratch"
        assert vgpr <= 256, f"{k}: {vgpr} VGPRs (two workgroups per CU need <= 256)"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_audit.py"), str(asm), "edge_kernel"],
                         check=True, capture_output=True, text=True).stdout
    counts = [int(x) for x in re.findall(r"hidden-load register hazards: (\d+)", out)]
    assert len(counts) == 5 and all(c == 0 for c in counts), out[-2000:]
