"""GPU box: a COLD build of the headline kernel's translation unit, on the target, from the sources in this tree.

The library the other tests load was built in the development container (hipcc cross-compiles gfx950) and travels with the
snapshot; vbmc_amd/build.py is incremental by mtime.  So that no kernel credit rests on a shipped binary alone, this test
  1. compiles vbmc_amd/csrc/ent_mfma_inst.hip with -DQS_VALUE=3 (D = 9, 10: k_entropy_mfma of the headline shape) from source
     with the box's own hipcc, keeping the assembly (--save-temps);
  2. reads the compiler's metadata of the two headline instantiations <QS=3, KT=3, grad, dense, one wave, tail 1, device RNG> -- the chunk
     grid and the walking launch: no AGPRs, no scratch (no spill), 2 waves per SIMD, the VGPR counts DESIGN.md section 4 quotes;
  3. links a library from that fresh object and the other objects, and runs the same evaluation through it (VBMC_HIP_LIB) and
     through the shipped library in two fresh processes: same sources, same compiler -> the same bits."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vbmc_amd", "csrc")
OBJ = os.path.join(ROOT, "vbmc_amd", "lib", "obj")

PROBE = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
import vbmc_amd
from bench import synth_inputs
D, N, K, S = 10, 60, 50, 3
inp = synth_inputs(0, D, N, K, S)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
th = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
th = np.asfortranarray(th[:, None] + 0.05 * np.random.default_rng(1).standard_normal((th.size, 3)))
o = vbmc_amd.negelcbo_batch(th, 0, vp, gp, 512, True, 0, seed=9)
print(json.dumps({"F": [x.hex() for x in o["F"]], "dF": float(np.sum(np.abs(o["dF"]))).hex(), "lib": vbmc_amd._lib.LIB_PATH}))
"""


@pytest.mark.timeout(900)
def test_headline_translation_unit_compiles_cold_on_the_target_and_runs(tmp_path):
    from vbmc_amd.build import FLAGS

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = str(tmp_path / "ent_mfma_qs3.o")
    r = subprocess.run([hipcc] + FLAGS + ["-Wno-pass-failed", "-DQS_VALUE=3", "--save-temps", "-c", os.path.join(CSRC, "ent_mfma_inst.hip"), "-o", obj],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = open(str(tmp_path / "ent_mfma_inst-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    meta = asm[asm.index("amdhsa.kernels:"):]
    # the two instantiations of the headline shape: the chunk grid and the walking launch (entropy_mfma.h: WALK) that serves wide batches
    for walk, label in ((0, "KT=3+tail+rng grad=1 sparse=0 HV=1:"), (1, "KT=3+tail+rng+walk grad=1 sparse=0 HV=1:")):
        found = None
        for body in re.split(r"\n  - \.agpr_count", "\n" + meta)[1:]:
            body = ".agpr_count" + body
            m = re.search(r"\.name:\s+(_Z14k_entropy_mfmaILi3ELi3ELb1ELb0ELi1ELi1ELb0ELb0ELb%dEEv7EntArgs)" % walk, body)
            if m:
                g = lambda k: int(re.search(k + r":\s+(\d+)", body).group(1))  # noqa: E731
                found = {"vgpr": g(r"\.vgpr_count"), "agpr": g(r"\.agpr_count"), "scratch": g(r"\.private_segment_fixed_size"),
                         "vgpr_spill": g(r"\.vgpr_spill_count"), "lds": g(r"\.group_segment_fixed_size")}
        assert found, "headline instantiation k_entropy_mfma<3,3,true,false,1,1,false,false,%s> not in the fresh object" % ("true" if walk else "false")
        assert found["vgpr"] <= 256 and found["agpr"] == 0 and found["scratch"] == 0 and found["vgpr_spill"] == 0, found
        # the figure DESIGN.md section 4 quotes is the one committed in profiles/isa_meta_qs3.txt (tools/isa_meta.py): the cold build must give it
        meta_line = [ln for ln in open(os.path.join(ROOT, "profiles", "isa_meta_qs3.txt")) if ln.startswith(label)][0]
        assert found["vgpr"] == int(re.search(r"vgpr (\d+)", meta_line).group(1)), (found, meta_line)
    # a library with the fresh object in place of the shipped one
    objs = [os.path.join(OBJ, "vbmc_hip.o")] + [obj if q == 3 else os.path.join(OBJ, "ent_mfma_qs%d.o" % q) for q in range(1, 10)]
    objs += [os.path.join(OBJ, "ent_lane_dt%d.o" % dt) for dt in (2, 4, 6, 8, 10, 12)]
    lib = str(tmp_path / "libvbmc_hip_cold.so")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    outs = []
    for env_lib in (lib, None):
        env = dict(os.environ)
        env.pop("VBMC_HIP_LIB", None)
        if env_lib:
            env["VBMC_HIP_LIB"] = env_lib
        r = subprocess.run([sys.executable, "-c", PROBE % ROOT], env=env, capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["lib"] == lib and outs[1]["lib"] != lib
    assert outs[0]["F"] == outs[1]["F"] and outs[0]["dF"] == outs[1]["dF"]
