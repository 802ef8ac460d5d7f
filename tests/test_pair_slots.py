"""The placement rules of the threaded node arrays below the top block (raytracer_amd/csrc/rtg_pair_slots.hpp) on the
host alone: a small C++ program builds random binary trees and checks that the slots the rules hand out are a
permutation of the pair slots with every follower behind its leader in one 128-byte line (tests/native/
pair_slots_check.cpp says what exactly).  No GPU and no HIP headers are needed: the header is plain C++."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pair_slots_are_a_permutation_with_followers_behind_their_leaders(tmp_path):
    exe = tmp_path / "pair_slots_check"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-o", str(exe),
                           os.path.join(ROOT, "tests", "native", "pair_slots_check.cpp")])
    out = subprocess.check_output([str(exe), "300"]).decode().split()
    assert out[0] == "ok", " ".join(out)
    assert int(out[1]) > 500 and int(out[2]) > 1_000_000
