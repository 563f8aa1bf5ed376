"""Stamp association (the producer of associate.txt) against the brute-force
restatement of the reference's rule, and through the files the input provider reads."""
import os

import numpy as np

from oracle import associate_oracle
from slam_amd.preprocess import associate as assoc


def _streams(n, seed, jitter=0.012, drop=0.1):
    rng = np.random.RandomState(seed)
    base = 1305031100.0 + np.arange(n) / 30.0
    first = {float(np.round(t, 6)): ["rgb/%.6f.png" % t] for t in base[rng.rand(n) > drop]}
    t2 = base + rng.uniform(-jitter, jitter, n)
    second = {float(np.round(t, 6)): ["depth/%.6f.png" % t] for t in t2[rng.rand(n) > drop]}
    return first, second


def test_associate_matches_brute_force():
    for seed, offset, max_diff in ((0, 0.0, 0.02), (1, 0.005, 0.02), (2, 0.0, 0.05), (3, -0.01, 0.011), (4, 0.0, 1e-9)):
        first, second = _streams(300, seed)
        got = assoc.associate(first, second, offset, max_diff)
        want = associate_oracle.associate(first, second, offset, max_diff)
        assert got == want
        assert len({a for a, _ in got}) == len(got) == len({b for _, b in got})   # each stamp used once
    assert assoc.associate({}, {1.0: ["x"]}, 0.0, 0.02) == []
    # a closer pair wins over an earlier one: 1.00 <-> 1.001, leaving 0.99 to 0.985
    got = assoc.associate({0.99: ["a"], 1.00: ["b"]}, {0.985: ["c"], 1.001: ["d"]}, 0.0, 0.02)
    assert got == [(0.99, 0.985), (1.00, 1.001)]


def test_read_file_list_and_association_text(tmp_path):
    seq = tmp_path / "freiburg1_xyz"
    os.makedirs(seq)
    (seq / "rgb.txt").write_text("# color images\n# timestamp filename\n"
                                 "1305031102.175304 rgb/1305031102.175304.png\n"
                                 "1305031102.211214,rgb/1305031102.211214.png\n"
                                 "1305031102.243211\trgb/1305031102.243211.png\n"
                                 "1305031102.999999\n\n")
    (seq / "depth.txt").write_text("# depth maps\n"
                                   "1305031102.160407 depth/1305031102.160407.png\n"
                                   "1305031102.226738 depth/1305031102.226738.png\n"
                                   "1305031102.262886 depth/1305031102.262886.png\n")
    rgb = assoc.read_file_list(str(seq / "rgb.txt"))
    assert sorted(rgb) == [1305031102.175304, 1305031102.211214, 1305031102.243211]   # stamp-only line skipped
    assert rgb[1305031102.211214] == ["rgb/1305031102.211214.png"]
    text = assoc.get_association(str(seq / "rgb.txt"), str(seq / "depth.txt"))
    assert text.splitlines() == [
        "1305031102.175304 rgb/1305031102.175304.png 1305031102.160407 depth/1305031102.160407.png",
        "1305031102.211214 rgb/1305031102.211214.png 1305031102.226738 depth/1305031102.226738.png",
        "1305031102.243211 rgb/1305031102.243211.png 1305031102.262886 depth/1305031102.262886.png"]
    assoc.create_association_data(str(tmp_path))
    rows = np.loadtxt(str(seq / "associate.txt"), dtype="str", unpack=False)     # as the input provider reads it
    assert rows.shape == (3, 4) and rows[0, 1].startswith("rgb") and rows[0, 3].startswith("depth")


def test_command_line(tmp_path, capsys):
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    a.write_text("1.00 x\n2.00 y\n")
    b.write_text("1.01 p\n2.50 q\n")
    assoc.main(["--first_file", str(a), "--second_file", str(b)])
    assert capsys.readouterr().out.splitlines() == ["1.000000 x 1.010000 p"]
    assoc.main(["--first_file", str(a), "--second_file", str(b), "--first_only", "--max_difference", "0.6"])
    assert capsys.readouterr().out.splitlines() == ["1.000000 x", "2.000000 y"]
