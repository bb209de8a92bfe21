"""Register / scratch budget of the kernels whose correctness was seen to depend on it (CPU: hipcc cross-compiles)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_prim_step_kernels_have_no_scratch(tmp_path):
    """The lazy Prim's step kernel pins its loaded values behind hand-placed waits (empty `asm volatile`s) and keeps ~136 registers at three workgroups
    a CU: it must compile without scratch.  (Round 5 held two multi-node variants of it to the same condition after one of them, spilling 12 bytes a lane,
    produced wrong trees for a reason nobody found; round 6 removed both variants -- measured no faster on cfg5's latent, opt-in -- rather than ship a kernel
    with an unexplained correctness condition.)"""
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
        for k in ("lazy_step_kernel",):
            if k in name and m:
                seen[k] = int(m.group(1))
    assert set(seen) == {"lazy_step_kernel"}, seen
    assert all(v == 0 for v in seen.values()), seen


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_plane_kernels_have_no_scratch_and_fit_their_cu(tmp_path):
    """The two-plane step's kernels (round 5): requests issued by inline asm keep their target registers through hand-placed waits -- a
    spilled register set would be reloaded from scratch before its request has landed.  No scratch, and the LDS / register budget of one
    512-thread workgroup per CU (dynamic LDS is set by the launchers: 128 KB and 144 KB + the tail's 4 KB of static LDS)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    want = {"planes.hip": ["l1_planes_kernel"], "wgrad_planes.hip": ["wgrad_dplanes_kernel"],
            "train_step.hip": ["wgrad_dplanes_rms_kernel", "wgrad_xplanes_rms_batched_kernel", "reduce_rms_kernel",
                               "mid_bwd_kernelILb0ELb1E", "mid_bwd_kernelILb0ELb0E"]}
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
            assert scratch == 0 and vgprs <= 256 and (lds <= 4096 or k.startswith("mid_bwd")), (k, scratch, vgprs, lds)      # (mid_bwd: a 1024-thread workgroup with 61 KB of static LDS)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_the_weight_gradient_operand_sets_are_out_of_the_compilers_sight(tmp_path):
    """Round 6 (wgrad_planes_device.h: dplanes_body): a K-step is one asm statement on FIXED operand sets v[200:223] / v[224:247]; the kernels that
    inline it carry amdgpu_num_vgpr(200).  In their ISA those registers appear only as the target of a `ds_read_b64_tr_b16` and as the A / B operand
    of a `v_mfma`; no MFMA reads a register a transposed read still has in flight (every statement ends with `s_waitcnt lgkmcnt(0)`); and nothing the
    compiler allocates reaches v200."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    reg = re.compile(r"\bv(\d+)\b|v\[(\d+):(\d+)\]")

    def regs(text):
        out = set()
        for m in reg.finditer(text):
            if m.group(1) is not None:
                out.add(int(m.group(1)))
            else:
                out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        return out

    for fn, kernels in (("wgrad_planes.hip", ["wgrad_dplanes_kernel"]), ("train_step.hip", ["wgrad_dplanes_rms_kernel", "wgrad_xplanes_rms_batched_kernel"])):
        src = os.path.join(ROOT, "idelucs_amd", "csrc", fn)
        out = tmp_path / (fn + ".s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", str(out)],
                           capture_output=True, text=True, cwd=os.path.dirname(src))
        assert r.returncode == 0, r.stderr[-2000:]
        cur, seen, flying = None, {}, set()
        for no, line in enumerate(open(out).read().split("\n")):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = next((k for k in kernels if k + "E" in m.group(1)), None)
                flying = set()
                continue
            if line.startswith(".Lfunc_end"):
                cur = None
            ins = line.split(";")[0].strip()
            if cur is None or not ins or ins.startswith("."):
                continue
            seen.setdefault(cur, [0, 0])
            ops = ins.split(None, 1)
            args = [a.strip() for a in ops[1].split(",")] if len(ops) > 1 else []
            if ops[0] == "ds_read_b64_tr_b16":
                d = regs(args[0])
                assert min(d) >= 200 and max(d) <= 247 and not (regs(args[1]) & set(range(200, 256))), (cur, no + 1, ins)
                flying |= d
                seen[cur][0] += 1
            elif ops[0] == "v_mfma_f32_32x32x16_f16":       # (the tail's fp32 MFMAs are the compiler's: they fall under the last rule)
                srcs = regs(args[1]) | regs(args[2])
                assert min(srcs) >= 200 and max(srcs) <= 247 and not (srcs & flying), (cur, no + 1, ins, sorted(srcs & flying))
                assert max(regs(args[0]) | regs(args[3])) < 200, (cur, no + 1, ins)
                seen[cur][1] += 1
            else:
                if ops[0] == "s_waitcnt" and "lgkmcnt(0)" in ins:
                    flying = set()
                assert not (regs(ins) & set(range(200, 256))), "%s: an operand set's register in `%s` (line %d of %s)" % (cur, ins, no + 1, out)
        assert set(seen) == set(kernels), (fn, seen)
        for k, (reads, mfmas) in seen.items():
            assert reads == 5 * 12 and mfmas == 4 * 6, (k, reads, mfmas)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_lds_read_leaves_a_hand_scheduled_k_step_in_flight(tmp_path):
    """VERDICT r5 #4a.  The rule of the hand-scheduled K-steps (l1_planes_device.h: L1P_STEP, wgrad_planes_device.h: WGD_STEP_*): a step is ONE asm statement from
    its first MFMA to the `s_waitcnt lgkmcnt(0)` that retires the LDS reads issued inside it, so no register is in flight at any point the compiler schedules
    around.  Read off the ISA by BASIC BLOCK (labels and branches cut them): in every block that holds an fp16 MFMA, LDS operations retire in order (`s_waitcnt
    lgkmcnt(N)` leaves the last N outstanding); no instruction touches the destination of a read that is still out, and no such block ends with a read outstanding."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    reg = re.compile(r"\bv(\d+)\b|v\[(\d+):(\d+)\]")

    def regs(text):
        out = set()
        for m in reg.finditer(text):
            if m.group(1) is not None:
                out.add(int(m.group(1)))
            else:
                out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        return out

    for fn, kernels in (("planes.hip", ["l1_planes_kernel"]), ("train_step.hip", ["l1_planes_batched_kernel", "wgrad_dplanes_rms_kernel", "wgrad_xplanes_rms_batched_kernel"]),
                        ("wgrad_planes.hip", ["wgrad_dplanes_kernel"])):
        src = os.path.join(ROOT, "idelucs_amd", "csrc", fn)
        out = tmp_path / (fn + ".s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", str(out)],
                           capture_output=True, text=True, cwd=os.path.dirname(src))
        assert r.returncode == 0, r.stderr[-2000:]
        # basic blocks of the kernels
        blocks, cur, blk = {}, None, None
        for no, line in enumerate(open(out).read().split("\n")):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = next((k for k in kernels if k + "E" in m.group(1)), None)
                blk = []
                if cur is not None:
                    blocks.setdefault(cur, []).append(blk)
                continue
            if line.startswith(".Lfunc_end"):
                cur = None
            if cur is None:
                continue
            if re.match(r"^\.LBB\w+:", line):
                blk = []
                blocks[cur].append(blk)
                continue
            ins = line.split(";")[0].strip()
            if not ins or ins.startswith("."):
                continue
            blk.append((no + 1, ins))
            if ins.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                blk = []
                blocks[cur].append(blk)
        assert set(blocks) == set(kernels), (fn, sorted(blocks))
        for k, bl in blocks.items():
            checked = 0
            for b in bl:
                if not any(i.startswith("v_mfma_f32_32x32x16_f16") for _, i in b):
                    continue
                fifo = []            # LDS operations outstanding, oldest first: the destination registers of a read, an empty set for a write / atomic
                for no, ins in b:
                    ops = ins.split(None, 1)
                    args = [a_.strip() for a_ in ops[1].split(",")] if len(ops) > 1 else []
                    flying = set().union(*fifo) if fifo else set()
                    if ops[0].startswith("ds_read"):
                        assert not (regs(args[1]) & flying), (k, no, ins)
                        fifo.append(regs(args[0]))
                        checked += 1
                    elif ops[0].startswith("ds_"):
                        assert not (regs(ins) & flying), (k, no, ins)
                        fifo.append(set())
                    elif ops[0] == "s_waitcnt":
                        m = re.search(r"lgkmcnt\((\d+)\)", ins)
                        if m:
                            n = int(m.group(1))
                            fifo = fifo[len(fifo) - n:] if n and n < len(fifo) else ([] if n == 0 else fifo)
                    else:
                        assert not (regs(ins) & flying), "%s: `%s` touches a register an LDS read has in flight (line %d of %s)" % (k, ins, no, out)
                assert not any(fifo), "%s: a block with fp16 MFMAs ends with LDS reads outstanding (line %d of %s)" % (k, b[-1][0], out)
            assert checked >= 24, (k, checked)


def test_the_one_developer_variable_is_parsed_by_name(tmp_path):
    """csrc/dev_env.h: IDELUCS_DEV="name=value,name=value" -- a key is matched whole (vec is not vec_ablate), a bare name reads as "1", a missing one as NULL,
    the value is re-read at every use; the Python side (idelucs_amd/_lib.py: DEV) splits the same string the same way."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    src = tmp_path / "t.cpp"
    src.write_text('#include <cstdio>\n#include "dev_env.h"\nint main(int argc, char **argv) { for (int i = 1; i < argc; ++i) { const char *v = idl::dev_env(argv[i]); '
                   'printf("%s=%s\\n", argv[i], v ? v : "<null>"); } return 0; }\n')
    exe = tmp_path / "t"
    r = subprocess.run([gxx, "-std=c++17", "-I", os.path.join(ROOT, "idelucs_amd", "csrc"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, IDELUCS_DEV="vec_ablate=3,vec=4,stamps,v3_ec=320,numa=off")
    out = subprocess.run([str(exe), "vec", "vec_ablate", "stamps", "v3_ec", "v3", "numa", "missing", "ec"], capture_output=True, text=True, env=env).stdout.split()
    assert out == ["vec=4", "vec_ablate=3", "stamps=1", "v3_ec=320", "v3=<null>", "numa=off", "missing=<null>", "ec=<null>"], out
    env.pop("IDELUCS_DEV")
    assert subprocess.run([str(exe), "vec"], capture_output=True, text=True, env=env).stdout.split() == ["vec=<null>"]
