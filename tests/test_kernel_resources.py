"""Register / scratch budget of the hot GEMM kernels: cross-compiles the GEMM sources to gfx950 assembly (no GPU needed, ~40 s)
and checks that
  * the tile-per-block forward/dgrad kernel stays at <= 128 VGPRs (4 waves per SIMD), the persistent stream-K variant and the
    weight-gradient kernel at <= 168 (3 waves per SIMD),
  * NO scratch (spill) instruction and no SGPR spill (v_readlane / v_writelane) sits inside a loop that carries MFMAs -- the spilled
    values of the forward kernel (tile bookkeeping) are written in the prologue and re-read in the epilogue.  (Round 5 rewrote the
    check: the round-4 version looked between the first and the last MFMA only and missed spill reloads at the TOP of the loop body),
  * the non-MFMA instruction count of the hot loop per 32 MFMAs stays at or below round 4's (73): every non-MFMA instruction a SIMD
    executes costs matrix-pipe time (profiles/r5_mfma_partner_probe.txt)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# name fragment -> (VGPR budget, scratch bytes budget, non-MFMA instructions per 32 MFMAs in the hot loop or None)
HOT = {
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELi0ELb0ELi0EEE": (128, 128, 76),      # <128,128,2,16,FAST,tile-per-block,fp32>
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELi1ELb0ELi0EEE": (168, 0, None),      # persistent stream-K (round 6: the hoisted deposit loop left no spill)
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELi2ELb0ELi0EEE": (128, 128, 76),      # tile-per-block + split-K tail in one launch (round 6)
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELi0ELb0ELi1EEE": (128, 128, 76),      # tile-per-block, ReLU epilogue records its bit mask
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELi1ELb0ELi1EEE": (168, 0, None),
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELi2ELb0ELi1EEE": (128, 128, 76),
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELi0ELb0ELi2EEE": (128, 128, 76),      # tile-per-block, epilogue masks with a recorded bit mask
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELi1ELb0ELi2EEE": (168, 0, None),
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELi2ELb0ELi2EEE": (128, 128, 76),
    "conv_wgradILi128ELi128ELi2ELb1ELb0ELb0ELb0EEE": (168, 64, None),
    "conv_wgradILi128ELi128ELi2ELb1ELb0ELb1ELb0EEE": (168, 64, None),        # QUAD: four pixels per lane (1x1 stride-1 layers)
    "conv_wgradILi128ELi128ELi2ELb1ELb0ELb1ELb1EEE": (168, 64, None),        # QUAD + QTAP: the same loader for "same"-padded k x k layers
}


def _asm(tmp_path, src):
    out = tmp_path / (src + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "da-sac_amd", "csrc"), "-S", "--cuda-device-only",
                           "-o", str(out), os.path.join(ROOT, "da-sac_amd", "csrc", src + ".hip")], stderr=subprocess.DEVNULL)
    return out.read_text()


def _loops(lines):
    """Loops of a kernel body as lists of lines, from LLVM's own block annotations ("Loop Header: Depth=" on the header block,
    "in Loop: Header=BBx_y" on every other block of the loop): [(lines of all blocks of the loop, #MFMA)].  (Round 6: the earlier
    version looked for a backward branch to a header label, which misses a loop whose latch is not its header and takes an
    out-of-line block that branches back into straight-line code for one.)"""
    groups, cur, inner = {}, None, set()
    for l in lines:
        m = re.match(r"^\.L(BB\d+_\d+):", l)
        if m:
            cur = None
            if "Loop Header" in l:
                cur = m.group(1)
                if "Inner Loop Header" in l:
                    inner.add(cur)
            else:
                h = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
                if h:
                    cur = h.group(1)
            if cur is not None:
                groups.setdefault(cur, [])
        if cur is not None:
            groups[cur].append(l)
    # innermost loops only: the persistent kernel's tile loop legitimately re-reads spilled scalars in its epilogue
    return [(body, sum("v_mfma" in x for x in body)) for h, body in groups.items() if h in inner]


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.isfile(HIPCC), reason="hipcc not available")
def test_hot_gemm_loops_have_no_scratch_fit_their_occupancy_and_keep_their_instruction_diet(tmp_path):
    txt = _asm(tmp_path, "conv_igemm")
    meta = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(
        r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size: (\d+)\n(?:.*\n){0,8}?\s+\.vgpr_count:\s+(\d+)", txt)}
    for key, (vgpr_budget, scratch_budget, diet) in HOT.items():
        names = [n for n in meta if key in n]
        assert len(names) == 1, (key, names)
        scratch, vgprs = meta[names[0]]
        assert vgprs <= vgpr_budget, (key, vgprs)
        assert scratch <= scratch_budget, (key, scratch)
        body = txt[txt.index("\n" + names[0] + ":"):]
        body = body[:body.index("s_endpgm")].split("\n")
        mfma = [i for i, l in enumerate(body) if "v_mfma" in l]
        assert len(mfma) >= 32
        loops = [lp for lp in _loops(body) if lp[1] >= 16]
        if loops:                                # every loop that carries MFMAs: no spill traffic of either kind
            for lbody, _ in loops:
                bad = [l.strip() for l in lbody if "scratch_" in l or "v_readlane" in l or "v_writelane" in l]
                assert not bad, (key, bad[:3])
        else:                                    # a fully unrolled K loop: no scratch traffic between its first and last MFMA
            bad = [l.strip() for l in body[mfma[0]:mfma[-1]] if "scratch_" in l]
            assert not bad, (key, bad[:3])
        if diet is not None:
            lbody, n = max(loops, key=lambda lp: lp[1])
            other = [l for l in lbody if l.strip() and not l.strip().startswith(";") and not re.match(r"^\.LBB", l.strip())
                     and "v_mfma" not in l]
            assert len(other) * 32.0 / n <= diet, (key, len(other), n)


# The streaming head kernels are issue-bound (tools/isa_count.py, profiles/EXPERIMENTS.md round 5): their occupancy arguments and
# instruction diets are part of the design.  name fragment -> (VGPR budget, scratch bytes budget, VALU instructions in the kernel)
HEAD = {
    "upsample_softmaxILi19ELb1ELb1EEE": (128, 32, 3400),      # probs path, up-factor 8: four waves per SIMD (was 225 VGPRs, ~4600 VALU)
    "upsample_softmaxILi19ELb0ELb1EEE": (64, 0, 1300),        # logits only: eight waves per SIMD (was 102 VGPRs, ~2500 VALU)
    "ce_bwd_rows_waveILi19ELi16EEE": (256, 0, 2800),          # two waves per SIMD, nothing spilled inside the row loop
}
POOL = {"maxpool_bwd_3s2p1": (64, 0, 200)}                    # 2 x 4 pixels per thread: ~170 VALU per 8 pixels (306 per 4 before)


@pytest.mark.parametrize("src,table", [("head", HEAD), ("pointwise", POOL)])
def test_streaming_head_kernels_keep_their_registers_and_instruction_counts(tmp_path, src, table):
    txt = _asm(tmp_path, src)
    meta = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(
        r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size: (\d+)\n(?:.*\n){0,8}?\s+\.vgpr_count:\s+(\d+)", txt)}
    for key, (vgpr_budget, scratch_budget, valu_budget) in table.items():
        names = [n for n in meta if key in n]
        assert len(names) == 1, (key, names)
        scratch, vgprs = meta[names[0]]
        assert vgprs <= vgpr_budget and scratch <= scratch_budget, (key, vgprs, scratch)
        body = txt[txt.index("\n" + names[0] + ":"):]
        body = body[:body.index("s_endpgm")].split("\n")
        valu = sum(1 for l in body if l.strip().startswith("v_"))
        assert valu <= valu_budget, (key, valu)
