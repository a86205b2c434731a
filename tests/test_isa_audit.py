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


@pytest.mark.timeout(600)
def test_edge16_kernels_keep_their_weights_in_accumulation_registers(tmp_path):
    """csrc/gw_edge16.hip: the persistent bf16 kernels are only fast while all weight registers stay in the AGPR half and feed
    the MFMAs from there.  Checked on the generated ISA of every instantiation (4 or 8 waves x
    residual from fp32 rows or bf16 tiles): no scratch, the register budget of its occupancy (512 for one wave per SIMD, 256
    for two), every MFMA takes its A operand from an AGPR, no AGPR <-> VGPR copies anywhere in the tile loop, 2 layers x
    4 groups x (64 / waves-per-matrix-quarter) MFMAs per tile, and an explicit wait state before and after each asm MFMA batch
    (inline asm is invisible to the hazard recogniser).  The layer-1 kernel (W_e in LDS) must fit two waves per SIMD unspilled."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "graph_weather_amd", "csrc", "gw_edge16.hip")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-c", src, "-o", "e.o", "-save-temps"]
    subprocess.run(cmd, cwd=tmp_path, check=True, capture_output=True)
    text = (tmp_path / "gw_edge16-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()

    def meta_of(name):
        meta = text[text.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        return (int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1)),
                int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)))

    names = re.findall(r"^(_Z\w*edge16_kernelILi\d+ELb[01]ELb[01]E\w*):", text, re.M)
    assert len(names) == 8, names  # waves 4 / 8 x residual rows / tiles x layer 1 from a workspace / gathered in the kernel
    for name in names:
        nw = int(re.search(r"edge16_kernelILi(\d+)E", name).group(1))
        scratch, vgpr = meta_of(name)
        # no scratch - except a couple of loop-invariant dwords the allocator parks in the 8-wave form that has BOTH the fp32-row
        # residual and the in-kernel gather (a combination only direct C-ABI callers reach: the forecaster hands the residual
        # over as bf16 tiles); they are reloaded at segment ends only
        allowed = 16 if ("ILi8ELb0ELb1E" in name) else 0
        assert scratch <= allowed, (name, scratch)
        assert vgpr <= (512 if nw == 4 else 256), (name, vgpr)
        body = text[text.index(name + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        lines = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]
        mfma = [ln for ln in lines if ln.startswith("v_mfma_f32_16x16x32_bf16")]
        per_group = 32 if nw == 4 else 16
        assert len(mfma) == 8 * per_group, (name, len(mfma))
        for ln in mfma:
            ops = [o.strip() for o in ln.split(None, 1)[1].split(",")]
            assert ops[1].startswith("a["), f"weight operand not in an AGPR: {ln}"
        first_barrier = next(i for i, ln in enumerate(lines) if ln.startswith("s_barrier"))
        loop = lines[first_barrier:]
        assert not any(ln.startswith(("v_accvgpr_read", "v_accvgpr_write")) for ln in loop), f"{name}: AGPR <-> VGPR copies in the tile loop"
        # wait states: an s_nop directly before the first and after the last MFMA of every batch (a batch = MFMAs with fewer
        # than 8 other instructions between neighbours - the scheduler drops waits and LDS reads between them)
        idx = [i for i, ln in enumerate(lines) if ln.startswith("v_mfma")]
        batches, cur = [], [idx[0]]
        for a_, b_ in zip(idx, idx[1:]):
            if b_ - a_ <= 8:
                cur.append(b_)
            else:
                batches.append(cur)
                cur = [b_]
        batches.append(cur)
        assert all(len(b) % 8 == 0 for b in batches), (name, [len(b) for b in batches])
        for b in batches:
            before = lines[max(0, b[0] - 6):b[0]]  # (instructions between the s_nop and the MFMA are wait states too)
            after = lines[b[-1] + 1:b[-1] + 12]
            assert any(ln.startswith("s_nop") for ln in before), f"{name}: no wait state before the MFMA batch at {b[0]}"
            assert any(ln.startswith("s_nop") for ln in after), f"{name}: no wait state after the MFMA batch at {b[-1]}"
    l1 = re.search(r"^(_Z\w*edge16_l1_kernel\w*):", text, re.M).group(1)
    scratch, vgpr = meta_of(l1)
    assert scratch == 0 and vgpr <= 256, (l1, scratch, vgpr)
    body = text[text.index(l1 + ":"):]
    body = body[:body.index(".end_amdhsa_kernel")]
    assert len(re.findall(r"^\s*v_mfma_f32_16x16x32_bf16", body, re.M)) == 128


@pytest.mark.timeout(600)
def test_team_kernels_feed_every_mfma_from_agprs_and_interleave_fillers(tmp_path):
    """csrc/gw_edge16t.hip (team-pipelined bf16 edge update): every instantiation keeps its matrix in the AGPR half (all 256
    MFMAs - 128 per team - read their A operand from an AGPR, no AGPR <-> VGPR copies after the prologue), fits two waves per SIMD
    (<= 256 registers) and stays near the register budget (the allocator parks a few loop-invariant dwords in scratch: bounded
    here - none of them is the destination of a load the kernel issues from asm, which are LDS-DMA only).  Inside the MFMA phases
    the stream is 'one filler per MFMA': no run of more than 12 non-MFMA instructions between two MFMAs of a group."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "graph_weather_amd", "csrc", "gw_edge16t.hip")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-c", src, "-o", "e.o", "-save-temps"]
    subprocess.run(cmd, cwd=tmp_path, check=True, capture_output=True)
    text = (tmp_path / "gw_edge16t-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
    names = re.findall(r"^(_Z\w*edge16t_kernelILb[01]ELb[01]ELb[01]ELb[01]E\w*):", text, re.M)
    # layer-1 tiles by DMA + residual tiles | gathered from fp32 rows | from fp16 rows (no residual) | the last two on segment-
    # aligned tiles (round 4: transposed output layer - the weight is then the B operand - and 8 segment-sum MFMAs on plain VGPRs)
    assert len(names) == 5, names
    for name in names:
        segt = "ELb1EEEv" in name  # edge16t_kernel<GATHER, PH, RES, SEGT = true>
        meta = text[text.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1))
        assert vgpr <= 256, (name, vgpr)
        assert scratch <= 256, (name, scratch)
        body = text[text.index(name + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        lines = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]
        # (the segment-aligned form with fp16 product rows runs its middle layer on fp16 operands: v_mfma_f32_16x16x32_f16)
        mfma = [i for i, ln in enumerate(lines) if ln.startswith(("v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x32_f16"))]
        assert len(mfma) == (264 if segt else 256), (name, len(mfma))
        layers = []
        for i in mfma:
            ops = [o.strip() for o in lines[i].split(None, 1)[1].split(",")]
            if ops[1].startswith("a[") or ops[2].startswith("a["):
                layers.append(i)
        assert len(layers) == 256, f"{name}: {len(layers)} MFMAs take their weight operand from an AGPR"
        if segt:  # team B's layer: activations as A (VGPR), weights as B (AGPR)
            assert sum(1 for i in layers if lines[i].split(None, 1)[1].split(",")[2].strip().startswith("a[")) == 128, name
            # LayerNorm sums: a 16-instruction reduce-scatter per group (8 row_mirror, 4 row_half_mirror, 2 + 2 quad_perm)
            dpp = [ln for ln in lines if ln.startswith("v_add_f32_dpp")]
            assert len(dpp) == 4 * 16 and sum("row_mirror" in ln for ln in dpp) == 32 and sum("row_half_mirror" in ln for ln in dpp) == 16 \
                and sum("quad_perm" in ln for ln in dpp) == 16, (name, len(dpp))
        first_barrier = next(i for i, ln in enumerate(lines) if ln.startswith("s_barrier"))
        assert not any(ln.startswith("v_accvgpr") for ln in lines[first_barrier:]), name
        # the phase clocks (s_memtime stamps + their branches) exist in tuning builds only: the tile loop is bound by what a wave
        # can issue between its MFMAs
        assert not any(ln.startswith("s_memtime") for ln in lines), name
        mfma = layers
        for phase in (mfma[:128], mfma[128:]):
            gaps = [b_ - a_ - 1 for a_, b_ in zip(phase, phase[1:])]
            assert max(gaps) <= (20 if segt else 12), (name, max(gaps))


@pytest.mark.timeout(600)
def test_processor_form_on_segment_tiles_keeps_weights_resident_and_has_no_scratch(tmp_path):
    """csrc/gw_edge16p.hip (processor block on segment-aligned tiles): both instantiations (e' written / dropped) fit two waves
    per SIMD without scratch, keep their matrix in the AGPR half (256 layer MFMAs read a weight operand from an AGPR - team A as
    the A operand, team B's transposed layer as the B operand - plus 8 segment-sum MFMAs on plain VGPRs), make no AGPR <-> VGPR
    copies after the prologue, reduce the LayerNorm statistics with a 16-instruction DPP reduce-scatter per group and stay
    within the LDS of one CU."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "graph_weather_amd", "csrc", "gw_edge16p.hip")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-c", src, "-o", "e.o", "-save-temps"]
    subprocess.run(cmd, cwd=tmp_path, check=True, capture_output=True)
    text = (tmp_path / "gw_edge16p-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
    names = re.findall(r"^(_Z\w*edge16p_kernelILb[01]E\w*):", text, re.M)
    assert len(names) == 2, names
    for name in names:
        meta = text[text.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1)) == 0, name
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)) <= 256, name
        body = text[text.index(name + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        lines = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]
        mfma = [ln for ln in lines if ln.startswith("v_mfma_f32_16x16x32_bf16")]
        assert len(mfma) == 264, (name, len(mfma))
        ops = [[o.strip() for o in ln.split(None, 1)[1].split(",")] for ln in mfma]
        assert sum(1 for o in ops if o[1].startswith("a[")) == 128, name   # middle layer: weight = A operand
        assert sum(1 for o in ops if o[2].startswith("a[")) == 128, name   # transposed output layer: weight = B operand
        first_barrier = next(i for i, ln in enumerate(lines) if ln.startswith("s_barrier"))
        assert not any(ln.startswith("v_accvgpr") for ln in lines[first_barrier:]), name
        dpp = [ln for ln in lines if ln.startswith("v_add_f32_dpp")]
        assert len(dpp) == 64 and sum("row_mirror" in ln for ln in dpp) == 32 and sum("quad_perm" in ln for ln in dpp) == 16, name


@pytest.mark.timeout(600)
def test_wide_kernels_have_no_scratch_and_the_gemm_keeps_three_waves_per_simd(tmp_path):
    """csrc/gw_wide.hip: the LayerNorm kernels hold a whole row per wave in registers (up to 64 columns per lane, three such arrays
    in the backward) - a spill would turn them into scratch-memory kernels; the GEMM's 128 x 128 tile must leave room for three
    workgroups per CU (LDS 40 KiB each, <= 168 registers per wave)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "graph_weather_amd", "csrc", "gw_wide.hip")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-c", src, "-o", "w.o", "-save-temps"]
    subprocess.run(cmd, cwd=tmp_path, check=True, capture_output=True)
    text = (tmp_path / "gw_wide-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
    seen = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, meta = m.group(1), m.group(2)
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1))
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", meta).group(1))
        assert scratch == 0, f"{name}: {scratch} bytes of scratch"
        assert vgpr <= 512, f"{name}: {vgpr} registers"
        seen[name] = (vgpr, lds)
    gemms = {k: v for k, v in seen.items() if "gemm_nt_kernel" in k}
    assert len(gemms) == 2, list(seen)
    for k, (vgpr, lds) in gemms.items():
        assert vgpr <= 168 and lds <= 40 * 1024 + 512, f"{k}: {vgpr} registers, {lds} bytes of LDS"
    for family, count in (("ln_fwd_wide_kernel", 4), ("ln_bwd_wide_kernel", 4), ("gather_sum_kernel", 1), ("segment_sum_wide_kernel", 1),
                          ("gather_wide_kernel", 1), ("relu_mask_wide_kernel", 1), ("add_rows_kernel", 1)):
        assert sum(family in k for k in seen) == count, (family, list(seen))


@pytest.mark.timeout(600)
def test_row_wise_kernels_keep_the_register_budget_of_two_workgroups_per_cu(tmp_path):
    """The 64-column forms of the bf16 row-wise kernel (csrc/gw_bf16.hip: 4 waves x 1 group; node update, + post products, + head,
    node encoder + post products) owe their speed to TWO workgroups per CU - one's row traffic under the other's MFMA phases -
    which holds only while a wave needs <= 256 registers and nothing spills; same for the fp32 chain kernels (two 64-column
    workgroups per CU by design) and the register-resident backward chain of round 3 (csrc/gw_kernels.hip)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")

    def kernels(fname):
        src = os.path.join(ROOT, "graph_weather_amd", "csrc", fname)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-c", src, "-o", "k.o", "-save-temps"]
        subprocess.run(cmd, cwd=tmp_path, check=True, capture_output=True)
        text = (tmp_path / (fname[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read_text()
        out = {}
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
            meta = m.group(2)
            out[m.group(1)] = (int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)),
                               int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1)))
        return out

    k16 = kernels("gw_bf16.hip")
    four_by_one = {n: v for n, v in k16.items() if "chain16_kernel" in n and n.endswith("ELi4ELi1EEEvN2gw9ChainArgsE")}
    assert len(four_by_one) == 4, sorted(four_by_one)  # node update | + post products | + head | mlp + post products
    for name, (vgpr, scratch) in four_by_one.items():
        assert vgpr <= 256 and scratch == 0, (name, vgpr, scratch)
    k32 = kernels("gw_kernels.hip")
    bwd = [v for n, v in k32.items() if "bwd_chain_kernel" in n]
    # (with / without the LayerNorm-backward prologue) x (with / without bias column sums and joined rows): ABI v19
    assert len(bwd) == 4 and all(v <= 256 and sc == 0 for v, sc in bwd), bwd
    chains = {n: v for n, v in k32.items() if "chain_kernel" in n and "bwd" not in n}
    assert len(chains) >= 8
    for name, (vgpr, scratch) in chains.items():
        # (the general 3-operand edge form parks 28 bytes of loop invariants: bounded, it is not on the forecaster's path)
        assert vgpr <= 256 and scratch <= 32, (name, vgpr, scratch)


@pytest.mark.timeout(600)
def test_split_kernels_fit_two_workgroups_per_cu_and_issue_three_mfmas_per_fragment_pair(tmp_path):
    """csrc/gw_split.hip (GW_DTYPE_BF16X3): every instantiation of the product build is the 4-wave x 1-group form, whose only
    latency hiding is the second workgroup on the CU - <= 256 registers, no scratch - and whose arithmetic is exactly three
    bf16 MFMAs (x_hi.w_hi, x_lo.w_hi, x_hi.w_lo) per pair of weight fragments read from LDS: MFMA count = 1.5 x the
    ds_read_b128 count of the passes, a multiple of 6 per unit.  No phase clocks in the product build."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "graph_weather_amd", "csrc", "gw_split.hip")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-c", src, "-o", "s.o", "-save-temps"]
    subprocess.run(cmd, cwd=tmp_path, check=True, capture_output=True)
    text = (tmp_path / "gw_split-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
    names = re.findall(r"^(_Z\w*14chainx3_kernel\w*):", text, re.M)
    assert len(names) == 11, names  # mlp x 5 shapes, edge, node update (+ post, + head), project, mlp + post
    assert all(n.endswith("ELi4ELi1ELi3EEEvN2gw9ChainArgsE") for n in names), names
    bwd = re.findall(r"^(_Z\w*18bwd_chainx3_kernel\w*):", text, re.M)  # the input-gradient chain of the training step (ABI v18)
    assert len(bwd) == 2, bwd  # with and without the LayerNorm-backward prologue (ABI v19)
    names = names + bwd
    assert "s_memtime" not in text
    # 256 x 256 passes per instantiation family: (raw layer-1 operands) + middle + output (+ products / head)
    for name in names:
        meta = text[text.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1))
        assert scratch == 0 and vgpr <= 256, (name, vgpr, scratch)
        body = text[text.index(name + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        lines = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]
        n_mfma = sum(ln.startswith("v_mfma_f32_16x16x32_bf16") for ln in lines)
        n_frag = sum(ln.startswith("ds_read_b128") for ln in lines)
        assert n_mfma > 0 and n_mfma % 3 == 0, (name, n_mfma)
        # a head's last layer reads fragments of 8 row tiles and multiplies 5 of them: at most 1.5 MFMAs per fragment read
        assert 2 * n_mfma <= 3 * n_frag, (name, n_mfma, n_frag)
        assert n_mfma >= 0.9 * 1.5 * n_frag, (name, n_mfma, n_frag)
        assert not any(ln.startswith(("v_mfma_f32_16x16x4", "v_mfma_f32_32x32")) for ln in lines), name


@pytest.mark.timeout(600)
def test_row_split_node_update_kernels_fit_three_waves_per_simd_without_scratch(tmp_path):
    """csrc/gw_noders.hip: a 12-wave workgroup (CG = 3) needs <= 168 registers per wave (three waves per SIMD) and no scratch in any
    instantiation - a spill would sit in front of the exchange / LayerNorm of every workgroup (round 6: a first bf16x3 form kept
    64 registers of fragments in flight and spilled 376 bytes per lane).  The matrix work is all there: fp32 6 passes x 64
    K-steps x 4 row tiles of straight-line code per wave, bf16x3 5 x 8 x 12 = 480 (counted below); and no scalar branch per MFMA group inside the passes
    (profiles/r05_x3_ablation.log: one branch per unit doubled a pass) - at most a handful per 64 KiB chunk."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "graph_weather_amd", "csrc", "gw_noders.hip")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-c", src, "-o", "n.o", "-save-temps"]
    subprocess.run(cmd, cwd=tmp_path, check=True, capture_output=True)
    text = (tmp_path / "gw_noders-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
    names = re.findall(r"^(_Z\w*node_rs3?_kernelILi\d+E\w*):", text, re.M)
    assert len(names) == 6, names  # fp32 / bf16x3 x 1, 2, 3 column groups per workgroup
    for name in names:
        meta = text[text.index(".amdhsa_kernel " + name):]
        meta = meta[:meta.index(".end_amdhsa_kernel")]
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1))
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", meta).group(1))
        cg = int(re.search(r"kernelILi(\d+)E", name).group(1))
        x3 = "node_rs3_kernel" in name
        assert scratch == 0, (name, scratch)
        assert vgpr <= (168 if cg == 3 else 256), (name, vgpr)
        assert lds == 0, (name, lds)  # dynamic LDS only (128 KiB at launch: two 64 KiB weight buffers)
        body = text[text.index(name + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")]
        lines = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]
        mfma = [ln for ln in lines if ln.startswith("v_mfma_f32_16x16x32_bf16" if x3 else "v_mfma_f32_16x16x4_f32")]
        # straight-line passes.  fp32: layer 1 with a raw node operand (2, the second refilling its registers) + layer 1 without (1)
        # + middle + output + the POST loop body = 6; bf16x3: node-operand pass + aggregate pass (shared by both modes) + 3 = 5
        assert len(mfma) == (5 * 96 if x3 else 6 * 256), (name, len(mfma))
        # branches between the first and the last MFMA: per-chunk partial DMA round (12 waves), pass hand-overs, operand-mode tests
        first = next(i for i, ln in enumerate(lines) if ln.startswith("v_mfma"))
        last = max(i for i, ln in enumerate(lines) if ln.startswith("v_mfma"))
        branches = sum(1 for ln in lines[first:last] if ln.startswith("s_cbranch"))
        # (the last chunk of a pass tests "is there a next pass" once per DMA round: 16 rounds with 4 waves, 6 with 12; a branch per
        # K-step / MFMA group in EVERY chunk would be >= 384 (fp32: 24 chunks x 16 K-steps) / >= 120 (bf16x3: 20 chunks x 6 groups))
        print(name, "VGPRs", vgpr, "branches between first and last MFMA", branches)
        assert branches <= (100 if x3 else 200), (name, branches)
