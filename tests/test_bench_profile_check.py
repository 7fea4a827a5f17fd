"""bench.py's use of the committed counter profiles (profiles/*_pmc_fetch_write.json): `roofline.traffic` is taken from a profile only
when the profile was made from the stage-kernel sources that are in the tree now -- tools/profile_bench.sh records their digest next
to the counters -- and refused, with the reason, otherwise.  No GPU: the arithmetic on a profile file."""
import argparse
import json
import os

import bench


def _args():
    return argparse.Namespace(workload="comp", default_size=True, contig=1_000_000, read_len=150, err_ppm=2000)


def _profile(tmp_path, monkeypatch, digest):
    prof = {"kg::k_p1v2_scatter<true, true, 512>": {"launches": 4, "FETCH_SIZE_KB_total": 1000.0, "WRITE_SIZE_KB_total": 3000.0},
            "kg::k_p2_fast<1, false, false>": {"launches": 8, "FETCH_SIZE_KB_total": 2000.0, "WRITE_SIZE_KB_total": 1000.0},
            "kg::k_synth_reads": {"launches": 1, "FETCH_SIZE_KB_total": 9e9, "WRITE_SIZE_KB_total": 9e9},           # not a count-stage kernel
            "_stage_sources_sha256_16": digest, "_stage_sources": bench.STAGE_SOURCES}
    p = tmp_path / "prof.json"
    p.write_text(json.dumps(prof))
    monkeypatch.setitem(bench.PROFILE_JSON, "comp", os.path.relpath(str(p), bench.ROOT))


def test_traffic_from_a_profile_of_these_sources(tmp_path, monkeypatch):
    _profile(tmp_path, monkeypatch, bench.stage_sources_digest())
    traffic, source = bench.pmc_traffic(_args(), 1)
    # (2 x FETCH + WRITE) of the count-stage kernels, KB -> bytes, per round (= launches of the level-1 scatter)
    assert traffic == int((2 * 1000 + 3000 + 2 * 2000 + 1000) * 1024 / 4)
    assert "4 rounds" in source


def test_a_profile_of_other_sources_is_refused(tmp_path, monkeypatch):
    _profile(tmp_path, monkeypatch, "0123456789abcdef")
    traffic, why = bench.pmc_traffic(_args(), 1)
    assert traffic is None and "other stage-kernel sources" in why and "profile_bench.sh" in why


def test_no_profile_for_other_sizes_or_several_gpus(tmp_path, monkeypatch):
    _profile(tmp_path, monkeypatch, bench.stage_sources_digest())
    a = _args()
    a.default_size = False
    assert bench.pmc_traffic(a, 1) == (None, None)
    assert bench.pmc_traffic(_args(), 2) == (None, None)


def test_the_digest_follows_the_stage_sources(tmp_path, monkeypatch):
    before = bench.stage_sources_digest()
    assert len(before) == 16 and all(os.path.exists(os.path.join(bench.ROOT, f)) for f in bench.STAGE_SOURCES)
    f = tmp_path / "k.hpp"
    f.write_text("// a kernel\n")
    monkeypatch.setattr(bench, "STAGE_SOURCES", bench.STAGE_SOURCES + [os.path.relpath(str(f), bench.ROOT)])
    assert bench.stage_sources_digest() != before
