"""The documents quote the committed profiles, not other runs: DESIGN.md's round-6 measurement table is generated from
profiles/r06_bench.json (tools/design_measurements.py) and must be identical to what the generator prints now; the round's
profiles describe the shipped code (no type that was deleted, no experiment switch left in the product source)."""
import glob
import importlib.util
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generator():
    spec = importlib.util.spec_from_file_location("design_measurements", os.path.join(ROOT, "tools", "design_measurements.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_design_quotes_the_committed_profile():
    g = _generator()
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert g.BEGIN in text and g.END in text
    have = text[text.index(g.BEGIN):text.index(g.END) + len(g.END)]
    assert have == g.block(), "DESIGN.md section 5 is stale: python3 tools/design_measurements.py --write"
    # ... and the headline figures in it are the compact line's (what the driver parses)
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_line.json")).read())
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    assert len(json.dumps(line)) < 4096 and line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert f"**{full['value']} Mpix/s**" in have and str(full["cfg5_balanced"]["value"]) in have and str(full["value_balanced"]) in have
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]


def test_the_rounds_profiles_describe_the_shipped_code():
    names = glob.glob(os.path.join(ROOT, "profiles", "r06_*"))
    assert names
    for path in names:
        if path.endswith(".patch"):
            continue
        assert "NodeRec32" not in open(path, errors="replace").read(), path      # (a type round 4 deleted after its trace was taken)
    # the kernel signatures in the round's traces are instantiations the shipped source has: nine template arguments
    for path in glob.glob(os.path.join(ROOT, "profiles", "r06_kernel_stats*.csv")):
        for sig in re.findall(r"k_eval_forest<([^>]*)>", open(path).read()):
            assert len(sig.split(",")) == 9, (path, sig)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "3d-beats_amd")):
        for f in files:
            if f.endswith((".hip", ".hpp", ".py", ".h")):
                assert "RDF_EXPERIMENT" not in open(os.path.join(dirpath, f)).read(), f
    assert "RDF_EXPERIMENT" not in open(os.path.join(ROOT, "include", "rdf_hip.h")).read()


def test_the_rounds_experiment_script_still_applies_to_the_sources():
    """profiles/r06_headline_ablation.txt was produced by variants that tools/make_exp_r06.py derives from a COPY of the product
    sources: its edits must still find their anchors in today's sources (no build here), and must leave the product untouched."""
    spec = importlib.util.spec_from_file_location("make_exp_r06", os.path.join(ROOT, "tools", "make_exp_r06.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    before = open(os.path.join(ROOT, "3d-beats_amd", "csrc", "rdf_hip.hip")).read()
    m.edit()
    assert open(os.path.join(ROOT, "3d-beats_amd", "csrc", "rdf_hip.hip")).read() == before
    edited = open(os.path.join(m.SRC, "rdf_hip.hip")).read()
    assert "RDF_ABL_PDF" in edited and "RDF_ABL_LDSNODES" in edited and "BLOCK == 896 ? 7" in edited and "RDF_ABL" not in before
