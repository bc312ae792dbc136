"""CPU check on the small-class entropy kernels (vbmc_amd/csrc/entropy_lane.h): every instantiation k_entropy_lane<DT, KP, grad>
compiles for gfx950 with NO spilled registers and no private segment.  A build of DT = 12, KP = 14 that spilled 87 registers
returned wrong, run-to-run different sums on the GPU (round 6); the register budget per instantiation (ENT_LANE_OCC) is what keeps
them at zero, and this test is what keeps that true."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vbmc_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _compile(dt, out):
    d = os.path.join(out, "dt%d" % dt)
    os.makedirs(d, exist_ok=True)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed", "-DDT_VALUE=%d" % dt,
                        "--save-temps=obj", "-c", os.path.join(CSRC, "ent_lane_inst.hip"), "-o", os.path.join(d, "l.o")],
                       capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(os.path.join(d, "ent_lane_inst-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def test_no_lane_kernel_spills(tmp_path):
    with ThreadPoolExecutor(max_workers=6) as ex:
        asms = list(ex.map(lambda dt: _compile(dt, str(tmp_path)), (2, 4, 6, 8, 10, 12)))
    seen = 0
    for asm in asms:
        for m in re.finditer(r"\.name:\s+(_Z14k_entropy_laneILi(\d+)ELi(\d+)ELb([01])EEv7EntArgs)\n(.*?)\.wavefront_size", asm, re.S):
            meta = m.group(5)
            spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
            priv = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1))
            vg = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
            assert spill == 0 and priv == 0, (m.group(2), m.group(3), m.group(4), spill, priv, vg)
            seen += 1
    assert seen == (4 * 8 + 5 + 4) * 2, seen          # DT = 2..8: KP = 2..16; DT = 10: KP <= 10; DT = 12: KP <= 8


def test_fused_prediction_kernel_does_not_spill(tmp_path):
    """k_pred_fused<QS, PT> (vbmc_amd/csrc/gp_kernels.h) for every (QS, PT) the host may launch -- PT <= PREDF_PT_FOR_QS(QS) -- compiles
    without spilled registers: the resident-tile count is chosen so (three tiles at D <= 12, two up to 20, one beyond)."""
    src = os.path.join(str(tmp_path), "pf.hip")
    combos = [(q, p) for q in range(1, 9) for p in range(1, (3 if q <= 3 else (2 if q <= 5 else 1)) + 1)]
    with open(src, "w") as f:
        f.write('#include "%s/common.h"\n#include "%s/device_math.h"\n#include "%s/trsm_mfma.h"\n#include "%s/gp_kernels.h"\n' % ((CSRC,) * 4))
        for q, p in combos:
            f.write("template __global__ void k_pred_fused<%d, %d>(PredArgs, const double*, const double*, const double*, double*, double*);\n" % (q, p))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-pass-failed", "-I" + os.path.join(ROOT, "include"),
                        "--save-temps=obj", "-c", src, "-o", os.path.join(str(tmp_path), "pf.o")], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    asm = open(os.path.join(str(tmp_path), "pf-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    seen = 0
    for m in re.finditer(r"\.name:\s+_Z12k_pred_fusedILi(\d+)ELi(\d+)EEv\S+\n(.*?)\.wavefront_size", asm, re.S):
        meta = m.group(3)
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1)) == 0, (m.group(1), m.group(2))
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1)) == 0, (m.group(1), m.group(2))
        seen += 1
    assert seen == len(combos), (seen, len(combos))
