"""The scheduled asm blocks of the leaf hash's row, lane and pair forms (csrc/*_asm.inc) are generated: each generator schedules its
blocks under the gfx950 wait-state rules, checks the rules on the result and executes every block with an interpreter of the
instructions used against the Poseidon round in Python integers.  Here: the generators pass their own checks, and the files
in the tree are what they produce (an edit by hand, or a generator changed without regenerating, fails)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


@pytest.mark.parametrize("tool,inc", [("gen_row_layer_asm.py", "row_layer_asm.inc"), ("gen_row_round_asm.py", "row_round_asm.inc"),
                                      ("gen_lane_round_asm.py", "lane_round_asm.inc"), ("gen_pair_round_asm.py", "pair_round_asm.inc")])
def test_generated_asm_is_current_and_self_checked(tool, inc):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]   # hazard rules and the interpreter's comparison are assertions inside
    have = open(os.path.join(ROOT, "starky_bls12_381_amd", "csrc", inc)).read()
    assert out.stdout == have
