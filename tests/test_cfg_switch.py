"""INTEGRATION.md section 2d, checked against the reference's own code: the cfg switch a maintainer would
add to falcon_kit/mains/consensus_task.py:70-104 (`falcon_sense_gpu = true` selects the GPU worker the way
`dazcon` selects its alternative, :92-96).  The patch text is taken FROM INTEGRATION.md, applied to the source
of the reference's module where it lies, and the function is run: the bash it emits must pipe LA4Falcon into
`python -m falcon_amd.mains.consensus` with the cfg's falcon_sense_option, keep the .tmp + mv convention, and
leave the script untouched when the key is absent.  Runs only where /root/reference is (the build container);
pypeflow is not installed there, so the two helpers the module takes from it are given as no-ops."""
import os
import re
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
TASK = os.path.join(REF, "falcon_kit", "mains", "consensus_task.py")

pytestmark = pytest.mark.skipif(not os.path.exists(TASK), reason="the reference is not on this box")


def _patch_text():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("### (d) A cfg switch in the task script"):]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "INTEGRATION.md section 2d lost its python block"
    return m.group(1)


def _load(patched):
    src = open(TASK).read()
    if patched:
        anchor = "    if config.get('dazcon', False):"
        assert src.count(anchor) == 1
        src = src.replace(anchor, _patch_text().rstrip("\n") + "\n\n" + anchor)
    # the module's relative imports: falcon_kit.io (pypeflow's helpers; only io.rm is used here) and .bash
    pkg = types.ModuleType("falcon_kit")
    pkg.__path__ = []
    io = types.ModuleType("falcon_kit.io")
    io.rm = lambda *a, **k: None
    bash = types.ModuleType("falcon_kit.bash")
    mains = types.ModuleType("falcon_kit.mains")
    mains.__path__ = []
    pkg.io, pkg.bash = io, bash
    saved = {k: sys.modules.get(k) for k in ("falcon_kit", "falcon_kit.io", "falcon_kit.bash", "falcon_kit.mains")}
    sys.modules.update({"falcon_kit": pkg, "falcon_kit.io": io, "falcon_kit.bash": bash, "falcon_kit.mains": mains})
    try:
        mod = types.ModuleType("falcon_kit.mains.consensus_task")
        mod.__package__ = "falcon_kit.mains"
        exec(compile(src, TASK, "exec"), mod.__dict__)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


CFG = dict(length_cutoff=12000, falcon_sense_skip_contained=False, falcon_sense_greedy=False, LA4Falcon_preload=False,
           falcon_sense_option="--output-multi --min-idt 0.70 --min-cov 4 --max-n-read 200 --n-core 6")   # fc_run_ecoli.cfg:33


def test_the_cfg_key_selects_the_gpu_worker():
    ref, new = _load(False), _load(True)
    args = ("raw_reads.db", "raw_reads.1.las", "cns_00001.fasta", 6)
    want = ref.script_run_consensus(dict(CFG), *args)
    # without the key (and with it false) the emitted script is the reference's, byte for byte
    assert new.script_run_consensus(dict(CFG), *args) == want
    assert new.script_run_consensus(dict(CFG, falcon_sense_gpu=False), *args) == want
    assert "python -m falcon_kit.mains.consensus --output-multi" in want
    got = new.script_run_consensus(dict(CFG, falcon_sense_gpu=True), *args)
    line = [ln for ln in got.splitlines() if "LA4Falcon" in ln]
    assert len(line) == 1
    line = line[0]
    # ... and is the reference's own pipe with the worker's module in it and the device binding in front
    # (the cfg's falcon_sense_option goes through the same get_falcon_sense_option as before)
    ref_line = [ln for ln in want.splitlines() if "LA4Falcon" in ln][0]
    assert "python -m falcon_kit.mains.consensus --output-multi --min-idt 0.70 --min-cov 4 --max-n-read 200" in ref_line
    assert line == "FALCON_AMD_DEVICES=${FALCON_AMD_DEVICES:-0} " + ref_line.replace("falcon_kit.mains.consensus",
                                                                                    "falcon_amd.mains.consensus")
    assert line.endswith(">| cns_00001.fasta.tmp") and "| python -m falcon_amd.mains.consensus " in line
    assert "falcon_kit.mains.consensus" not in got
    # the rest of the task script is the reference's: pipefail, the cut-off, .tmp + mv
    assert got.replace(line, "") == want.replace(ref_line, "")
    # dazcon still wins when both keys are set, as in the reference
    assert "dazcon" in new.script_run_consensus(dict(CFG, falcon_sense_gpu=True, dazcon=True), *args)


def test_the_module_the_script_names_is_the_worker():
    """`python -m falcon_amd.mains.consensus` with the repo on PYTHONPATH is the worker's command line: same
    flags as the reference's (`--help` lists every one the cfgs use), no GPU needed to ask."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-m", "falcon_amd.mains.consensus", "--help"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=120)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    for flag in ("--output-multi", "--min-idt", "--min-cov", "--max-n-read", "--n-core", "--trim", "--output-full"):
        assert flag in out, flag
