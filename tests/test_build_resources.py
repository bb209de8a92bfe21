"""Register / scratch budget of the kernels whose correctness was seen to depend on it (CPU: hipcc cross-compiles)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_prim_step_kernels_have_no_scratch(tmp_path):
    """Round 5: a build of lazy_fold_kernel that spilled 12 bytes a lane to scratch (168 registers at three workgroups a CU)
    produced wrong records -- trees that differ from sklearn's from the second edge on; the same source without the spill is exact.
    Not understood (the step kernels pin loaded values with empty `asm volatile`s), so the condition is held by a test: the three
    step kernels of idl_mst_prim_lazy compile without scratch."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "idelucs_amd", "csrc", "mst.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", src, "-o", str(tmp_path / "mst.o"),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=os.path.dirname(src))
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    seen = {}
    for b in blocks:
        name = b.split()[0]
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b)
        for k in ("lazy_step_kernel", "lazy_fold_kernel", "lazy_multi_kernel"):
            if k in name and m:
                seen[k] = int(m.group(1))
    assert set(seen) == {"lazy_step_kernel", "lazy_fold_kernel", "lazy_multi_kernel"}, seen
    assert all(v == 0 for v in seen.values()), seen


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_plane_kernels_have_no_scratch_and_fit_their_cu(tmp_path):
    """The two-plane step's kernels (round 5): requests issued by inline asm keep their target registers through hand-placed waits -- a
    spilled register set would be reloaded from scratch before its request has landed.  No scratch, and the LDS / register budget of one
    512-thread workgroup per CU (dynamic LDS is set by the launchers: 128 KB and 144 KB + the tail's 4 KB of static LDS)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    want = {"planes.hip": ["l1_planes_kernel"], "wgrad_planes.hip": ["wgrad_xplanes_kernel"],
            "train_step.hip": ["wgrad_xplanes_rms_kernel", "reduce_rms_kernel", "l1p_rms_kernel"]}
    for fn, kernels in want.items():
        src = os.path.join(ROOT, "idelucs_amd", "csrc", fn)
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", src, "-o", str(tmp_path / (fn + ".o")),
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=os.path.dirname(src))
        assert r.returncode == 0, r.stderr[-2000:]
        blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
        seen = {}
        for b in blocks:
            name = b.split()[0]
            for k in kernels:
                if k in name:
                    seen[k] = (int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)), int(re.search(r" VGPRs: (\d+)", b).group(1)),
                               int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1)))
        assert set(seen) == set(kernels), (fn, seen)
        for k, (scratch, vgprs, lds) in seen.items():
            assert scratch == 0 and vgprs <= 256 and lds <= 4096, (k, scratch, vgprs, lds)
