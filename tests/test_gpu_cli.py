"""The real command line on the GPU: stdin -> stdout byte parity with the golden
vectors produced by the reference's own driver (tests/golden/f5_cli, f6_cli_trim),
through the module path, the drop-in ``falcon_kit`` overlay and the console script."""
import os
import subprocess
import sys

import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F5 = load_golden("f5_cli")
F6 = load_golden("f6_cli_trim")


def run_cmd(cmd, stdin_text, extra_path=None):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join(([extra_path] if extra_path else []) + [ROOT])
    p = subprocess.run(cmd, input=stdin_text, capture_output=True, text=True, env=env, cwd=ROOT,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout


@pytest.mark.parametrize("case", F5["runs"] + F6["runs"],
                         ids=[" ".join(r["argv"]) or "defaults" for r in F5["runs"] + F6["runs"]])
def test_module_cli(case):
    out = run_cmd([sys.executable, "-m", "falcon_amd.mains.consensus"] + case["argv"] +
                  ["--n-core", "1"], F5["stdin"])
    assert out == case["stdout"]


def test_dropin_overlay_and_console_script():
    case = F5["runs"][1]  # the fc_run_ecoli.cfg flags
    out = run_cmd([sys.executable, "-m", "falcon_kit.mains.consensus"] + case["argv"] +
                  ["--n-core", "1"], F5["stdin"], extra_path=os.path.join(ROOT, "dropin"))
    assert out == case["stdout"]
    out = run_cmd([os.path.join(ROOT, "bin", "fc_consensus")] + case["argv"] + ["--n-core", "1"],
                  F5["stdin"])
    assert out == case["stdout"]


def test_overlay_exposes_reference_names():
    code = ("import falcon_kit, sys; from falcon_kit import kup, DWA, falcon, ConsensusData;"
            "import falcon_kit.falcon_kit as fk;"
            "print(fk.consensus_of(['ACGT'*300]*12, 4, 8, 0.7)[:20])")
    out = run_cmd([sys.executable, "-c", code], "", extra_path=os.path.join(ROOT, "dropin"))
    assert out.strip() == ("ACGT" * 300)[1:21]
