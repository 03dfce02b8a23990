"""The GPU-side scripts run where a typo costs box minutes: what scripts/evidence.sh (and the A/B helpers) name must
exist -- the scripts they call, the bench.py options and workloads they pass, the python modules they import."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = ["evidence.sh", "ab_times.sh", "ab_counters.sh", "check_align2.sh", "build_variant.sh"]


def _text(name):
    return open(os.path.join(ROOT, "scripts", name)).read()


def test_shell_syntax():
    for s in SCRIPTS:
        subprocess.run(["bash", "-n", os.path.join(ROOT, "scripts", s)], check=True)


def test_every_script_and_option_they_name_exists():
    sys.path.insert(0, ROOT)
    import bench
    known = set()
    ap_src = open(os.path.join(ROOT, "bench.py")).read()
    known.update(re.findall(r'add_argument\("(--[a-z0-9-]+)"', ap_src))
    workloads = set(bench.WORKLOADS) | {"trim", "align1500", "utg", "e2e-long"}
    for s in SCRIPTS:
        text = _text(s)
        for path in set(re.findall(r"\$R/(scripts/[\w/.]+\.(?:py|sh))", text)):
            assert os.path.exists(os.path.join(ROOT, path)), (s, path)
        for line in text.splitlines():
            if "bench.py" not in line:
                continue
            after = line.split("bench.py", 1)[1]
            for opt in re.findall(r"(?<![\w-])(--[a-z][a-z0-9-]+)", after):
                if opt in ("--kernel-trace", "--stats", "--pmc", "--kernel-include-regex", "--output-format"):
                    continue  # (rocprofv3's own, in front of `--`)
                assert opt in known, (s, opt, line.strip()[:120])
            for w in re.findall(r"--workload (\$?\w[\w-]*)", after):
                assert w.startswith("$") or w in workloads, (s, w)
    # the loop variables of evidence.sh's workload loops
    for group in re.findall(r"for w in ([a-z0-9 -]+); do", _text("evidence.sh")):
        for w in group.split():
            assert w in workloads, w


def test_the_record_scripts_import():
    for mod in ("pmc_table", "pmc_issue_record", "pmc_traffic_record", "rocpd_summary", "a2_bytes"):
        src = open(os.path.join(ROOT, "scripts", mod + ".py")).read()
        compile(src, mod, "exec")
