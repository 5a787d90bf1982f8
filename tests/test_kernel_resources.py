"""Register / scratch budget of the four hot GEMM instantiations (VERDICT r1 #6b): cross-compiles conv_igemm.hip to gfx950
assembly (no GPU needed, ~10 s) and checks that
  * the tile-per-block forward/dgrad kernel stays at <= 128 VGPRs (4 waves per SIMD) and the persistent stream-K variant and
    the weight-gradient kernel at <= 168 (3 waves per SIMD),
  * NO scratch (spill) instruction sits between the first and the last MFMA of any of them -- i.e. inside the K loop; the few
    spilled values of the forward kernel (tile-index bookkeeping) are written in the prologue and re-read in the epilogue,
    those of the stream-K variant sit around the tile hand-off."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOT = {
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELb0ELb0ELi0EEE": 128,      # <128,128,2,16,FAST,tile-per-block,fp32>
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELb1ELb0ELi0EEE": 168,      # stream-K
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELb0ELb0ELi1EEE": 128,      # tile-per-block, ReLU epilogue records its bit mask
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELb1ELb0ELi1EEE": 168,
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELb0ELb0ELi2EEE": 128,      # tile-per-block, epilogue masks with a recorded bit mask
    "conv_gemmILi128ELi128ELi2ELi16ELb1ELb1ELb0ELi2EEE": 168,
    "conv_wgradILi128ELi128ELi2ELb1ELb0ELb0EEE": 168,
    "conv_wgradILi128ELi128ELi2ELb1ELb0ELb1EEE": 168,        # QUAD: four pixels per lane (1x1 stride-1 layers)
}


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.isfile(HIPCC), reason="hipcc not available")
def test_hot_gemm_loops_have_no_scratch_and_fit_their_occupancy(tmp_path):
    out = tmp_path / "conv_igemm.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "da-sac_amd", "csrc"), "-S", "--cuda-device-only",
                           "-o", str(out), os.path.join(ROOT, "da-sac_amd", "csrc", "conv_igemm.hip")], stderr=subprocess.DEVNULL)
    txt = out.read_text()
    meta = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(
        r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size: (\d+)\n(?:.*\n){0,8}?\s+\.vgpr_count:\s+(\d+)", txt)}
    for key, budget in HOT.items():
        names = [n for n in meta if key in n]
        assert len(names) == 1, (key, names)
        scratch, vgprs = meta[names[0]]
        assert vgprs <= budget, (key, vgprs)
        body = txt[txt.index("\n" + names[0] + ":"):]
        body = body[:body.index("s_endpgm")].split("\n")
        mfma = [i for i, l in enumerate(body) if "v_mfma" in l]
        assert len(mfma) >= 32
        inside = [l.strip() for l in body[mfma[0]:mfma[-1]] if "scratch_" in l]
        assert not inside, (key, inside[:3])
        # bytes of scratch per lane.  The persistent stream-K variant may park up to 48 values around the tile hand-off (once per
        # tile cut by a range boundary: half a deposit, 32 registers, is in flight next to the 64 accumulators); everything else
        # keeps the round-1 budget of a few prologue values.
        assert scratch <= (192 if "ELb1ELb1ELb0ELi" in key else 64), (key, scratch)
