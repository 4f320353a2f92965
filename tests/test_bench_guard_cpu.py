"""bench.py's regression guard (round 6): the round-5 line against round 4's must name the adversarial record (5.59 vs 47.6 volumes/s went unnoticed through
four committed lines); a line against itself names nothing."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.readline())


def test_round5_line_against_round4_flags_the_adversarial_record():
    rep = bench.regression_report(_line("r05_final_bench_line.json"), ref_path=os.path.join(ROOT, "profiles", "r04_final_bench_line.json"))
    names = [r["record"] for r in rep["regressions"]]
    assert "adversarial.value" in names, rep
    assert "value" not in names and "secondary.value" not in names, rep      # the headline numbers improved
    adv = next(r for r in rep["regressions"] if r["record"] == "adversarial.value")
    assert adv["ratio"] < 0.2


def test_a_line_against_itself_is_clean_and_lower_is_better_records_are_read_the_right_way():
    line = _line("r05_final_bench_line.json")
    rep = bench.regression_report(line, ref_path=os.path.join(ROOT, "profiles", "r05_final_bench_line.json"))
    assert rep["regressions"] == [] and rep["compared"]["value"] == 1.0
    slow = json.loads(json.dumps(line))
    slow["end_to_end"]["total_s"] *= 1.5
    slow["inference"]["value"] *= 1.5
    rep = bench.regression_report(slow, ref_path=os.path.join(ROOT, "profiles", "r05_final_bench_line.json"))
    assert [r["record"] for r in rep["regressions"]] == ["end_to_end.total_s"]


def test_default_reference_is_the_newest_committed_final_line():
    rep = bench.regression_report(_line("r05_final_bench_line.json"))
    assert rep["against"].startswith("profiles/r") and rep["against"].endswith("_final_bench_line.json")
