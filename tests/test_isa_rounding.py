"""The forest kernels switch the wave's fp32 rounding mode around the level loop (rdf_device.hpp: set_round_down /
set_round_nearest / pin); the compiler knows nothing of it and may move an fp instruction across a switch.  This test
reads the compiled gfx950 ISA (tools/check_rounding_isa.py: control-flow graph + mode propagation over every k_eval_forest
instantiation) and refuses an instruction that every path reaches in the wrong mode: an add / mul / divide sequence in
round-down mode, a packed fma (the one-fma divide-and-floor) in round-to-nearest.  Runs on the CPU: hipcc cross-compiles."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_fp_instruction_sits_in_the_wrong_rounding_mode():
    import check_rounding_isa
    rows = check_rounding_isa.report()
    assert len(rows) >= 40                                     # every instantiation was found and parsed
    headline = [r for r in rows if r[0] == "k_eval_forest<512, true, 4, false, 4, false, 1, false, false>"]
    assert headline and headline[0][3] > 20                   # ... and the analysis saw its round-down region
    bad = {short: definite for short, definite, _, _, _ in rows if definite}
    assert not bad, bad
