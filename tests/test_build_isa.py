"""The generated gfx950 code of the MFMA kernels, checked on the CPU (hipcc cross-compiles without a GPU).

Round 2 found two ways in which hipcc (ROCm 7.2) miscompiles chains of v_mfma_f64_16x16x4_f64, round 6 a third:
  * the destination registers of an MFMA with a constant-zero accumulator may overlap a dying A / B source register;
  * the wait states between an MFMA and a VALU read of its result are missing in some instantiations (the reverse sweep
    read the last destination pair one slot after the MFMA and got the accumulator from before the last k-step);
  * gfx950 does not interlock "VALU writes a VGPR -> an MFMA reads it as a source" (three issue slots are needed, measured by
    tools/ubench_srcc_war.hip), and hipcc pads only behind its OWN VALU instructions: behind a VALU instruction inside an
    inline-asm statement the MFMA may come too early and read the old register (the table exp's exponent insertion in front
    of the reverse sweep's moment product: wrong, run-to-run different sums in whichever instantiation the scheduler put
    the two within two slots of each other -- round 5's unexplained K = 8 failure).
csrc/mm_device.h carries the source-level counter-measures (MFMA_KEEP_ALIVE, MFMA_RESULT_FENCE); this test compiles the
MFMA-carrying translation units to assembly and scans EVERY instantiation with tools/mfma_overlap_check.py and
tools/mfma_hazard_check.py, so a compiler or source change that re-opens either hole fails here, not on the GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


UNITS = ["pair", "bwd", "linalg", "fitc_train", "prep_dt_a", "prep_dt_b"]   # prep_dt_a, _b: the heads for D <= 11 (both register builds), whose one-launch small step carries the pair and sweep arithmetic


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    """All units compiled to assembly ONCE, side by side (the scan of a unit takes seconds, its compilation a minute or two)."""
    out = tmp_path_factory.mktemp("isa")
    procs = {}
    for unit in UNITS:
        src = os.path.join(ROOT, "pilco_amd", "csrc", unit + ".hip")
        asm = str(out / (unit + ".s"))
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
               "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "-o", asm, src]
        procs[unit] = (subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True), asm)
    res = {}
    for unit, (pr, asm) in procs.items():
        try:
            _, err = pr.communicate(timeout=1500)
        except subprocess.TimeoutExpired:
            pr.kill()
            _, err = pr.communicate()
        res[unit] = (pr.returncode, asm, err)
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("unit", UNITS)
def test_mfma_chains_have_no_register_overlap_and_no_early_result_reads(unit, assembly):
    rc, asm, err = assembly[unit]
    assert rc == 0 and os.path.exists(asm), err[-2000:]
    assert "v_mfma_f64_16x16x4_f64" in open(asm).read()
    for tool in ("mfma_overlap_check.py", "mfma_hazard_check.py"):
        chk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), asm], capture_output=True, text=True, timeout=300)
        assert chk.returncode == 0, "%s on %s:\n%s" % (tool, unit, chk.stdout[-3000:])


def test_the_scanners_flag_the_two_patterns_they_guard_against(tmp_path):
    """The code hipcc actually emitted in round 2 (reverse sweep, K = 20 instantiation), reduced to the offending lines."""
    tools = os.path.join(ROOT, "tools")
    overlap = tmp_path / "overlap.s"
    overlap.write_text("_Zbad:\n\tv_mfma_f64_16x16x4_f64 v[108:115], v[114:115], v[80:81], 0\n")
    early = tmp_path / "early.s"
    early.write_text("_Zbad:\n\tv_mfma_f64_16x16x4_f64 v[108:115], v[178:179], v[96:97], v[108:115]\n"
                     "\tv_max_f64 v[194:195], v[114:115], s[68:69]\n")
    fine = tmp_path / "fine.s"
    fine.write_text("_Zok:\n\tv_mfma_f64_16x16x4_f64 v[108:115], v[178:179], v[96:97], v[108:115]\n\ts_nop 10\n"
                    "\tv_max_f64 v[194:195], v[114:115], s[68:69]\n")
    # round 6: an inline-asm VALU write one instruction in front of the MFMA that reads the register (hipcc's code for the sweep)
    war = tmp_path / "asmw.s"
    war.write_text("_Zbad:\n\t;;#ASMSTART\n\tv_lshl_add_u32 v91, v42, 12, v91\n\t;;#ASMEND\n\tv_mov_b64_e32 v[48:49], v[40:41]\n"
                   "\tv_mfma_f64_16x16x4_f64 v[18:25], v[90:91], v[140:141], v[18:25]\n\ts_nop 10\n\ts_nop 10\n")
    war_ok = tmp_path / "asmw_ok.s"   # ... and with the wait states in between
    war_ok.write_text("_Zok:\n\t;;#ASMSTART\n\tv_lshl_add_u32 v91, v42, 12, v91\n\t;;#ASMEND\n\ts_nop 1\n"
                      "\tv_mfma_f64_16x16x4_f64 v[18:25], v[90:91], v[140:141], v[18:25]\n\ts_nop 10\n\ts_nop 10\n")
    run = lambda tool, f: subprocess.run([sys.executable, os.path.join(tools, tool), str(f)], capture_output=True, text=True).returncode
    assert run("mfma_hazard_check.py", war) == 1 and run("mfma_hazard_check.py", war_ok) == 0
    assert run("mfma_overlap_check.py", overlap) == 1
    assert run("mfma_hazard_check.py", early) == 1
    assert run("mfma_overlap_check.py", fine) == 0 and run("mfma_hazard_check.py", fine) == 0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("unit", UNITS)
def test_kernarg_warm_stays_inside_the_kernel_argument_segment(unit, assembly):
    """csrc/common.h: kernarg_warm<BYTES> requests every 64-byte line of the kernel-argument segment with one s_load_dword each
    (one scalar-cache round trip instead of one per field) -- up to a hand-written byte count.  The count must not run past
    the segment the code object declares (.kernarg_segment_size: explicit + hidden arguments), whatever the code-object
    version or a changed argument list make of it."""
    import re
    rc, asm, err = assembly[unit]
    assert rc == 0, err[-2000:]
    text = open(asm).read()
    seg = {m.group(2): int(m.group(1)) for m in re.finditer(r"\.kernarg_segment_size:\s*(\d+)(?:.|\n)*?\.name:\s*(\S+)", text)}
    if not seg:   # (metadata order: .name before .kernarg_segment_size in some versions)
        seg = {m.group(1): int(m.group(2)) for m in re.finditer(r"\.name:\s*(\S+)(?:.|\n)*?\.kernarg_segment_size:\s*(\d+)", text)}
    assert seg, "no kernel metadata found in " + asm
    func, inasm, worst, checked, skip = None, False, {}, 0, False
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            func = m.group(1)
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
        elif t.startswith(";;#ASMEND"):
            inasm = False
        elif inasm and t.startswith(".if"):          # the assembler's own conditionals around the lines past the segment
            c = re.match(r"\.if\s+(0x[0-9a-fA-F]+|\d+)\s*>\s*(0x[0-9a-fA-F]+|\d+)", t)
            skip = bool(c) and not int(c.group(1), 0) > int(c.group(2), 0)
        elif inasm and t.startswith(".endif"):
            skip = False
        elif inasm and func in seg and not skip:
            m = re.match(r"s_load_dword\s+s\d+,\s*s\[\d+:\d+\],\s*(0x[0-9a-fA-F]+|\d+)", t)
            if m:
                worst[func] = max(worst.get(func, 0), int(m.group(1), 0) + 4)
    for f, end in worst.items():
        checked += 1
        assert end <= seg[f], "%s: kernarg_warm reads up to byte %d of a %d-byte kernel-argument segment" % (f, end, seg[f])
    if unit in ("pair", "bwd", "linalg", "prep_dt_a"):
        assert checked > 0, "no kernarg_warm found in " + unit
