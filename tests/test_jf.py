""".jf (Jellyfish binary/sorted) interop, host side -- no GPU needed.

The reader/writer in kat_amd/csrc/kg_jf.cpp are checked against the reference's own fixture tests/data/ecoli.header.jf27
and the assertions of the reference's tests/check_jellyfish.cc (header fields :38-60, record count :93-116, dump ->
reload round trip :158-180), and against the oracle's independent reader."""
import json
import os

import numpy as np
import pytest

import kat_amd


def read_header(path):
    raw = open(path, "rb").read()
    hlen = int(raw[:9])
    return json.loads(raw[9:9 + hlen].rstrip(b"\0")), 9 + hlen, raw


def matrix_times(cols, key):
    c, res = len(cols), 0
    for i in range(c):
        if key >> i & 1:
            res ^= cols[c - 1 - i]
    return res


def test_reference_fixture(ko, refdata):
    p = os.path.join(refdata, "ecoli.header.jf27")
    k, canonical, keys, counts = kat_amd.jf_read_records(p)
    assert (k, canonical, keys.size) == (27, False, 1889)                                      # check_jellyfish.cc:50,115
    ok_, oc = ko.Table.from_jf(p).dump_sorted()
    o = np.argsort(keys)
    assert np.array_equal(keys[o], ok_) and np.array_equal(counts[o], oc)
    q = {ko.encode(s): c for s, c in (("AGCTTTTCATTCTGACTGCAACGGGCA", 3), ("GCATAGCGCACAGACAGATAAAAATTA", 1),
                                       ("AATGAAAAAGGCGAACTGGTGGTGCTT", 1), ("CTCACCAATGTACATGGCCTTAATCTG", 1))}
    got = dict(zip(keys.tolist(), counts.tolist()))
    assert all(got[key] == c for key, c in q.items())                                           # check_jellyfish.cc:82-85
    # the fixture itself obeys the order the writer reproduces: (M * kmer) & (size - 1), then kmer
    hdr, off, raw = read_header(p)
    cols, size = hdr["matrix1"]["columns"], hdr["size"]
    order = [(matrix_times(cols, int(key)) & (size - 1), int(key)) for key in keys]
    assert order == sorted(order) and off == 1368                                               # check_jellyfish.cc:54


def test_write_then_reload_roundtrip(ko, refdata, tmp_path):
    """check_jellyfish.cc:158-180 (dump, reload, same answers) + structural checks of what was written."""
    rng = np.random.default_rng(3)
    for k, canonical, n in ((27, True, 5000), (31, False, 1), (32, False, 300), (5, True, 0), (13, False, 2000)):
        keys = rng.choice(4 ** min(k, 31), size=n, replace=False).astype(np.uint64) if n else np.zeros(0, np.uint64)
        if k == 32 and n:
            keys[0] = np.uint64(2 ** 64 - 1)
        counts = rng.integers(1, 1000, size=n).astype(np.uint64)
        if n > 2:
            counts[1] = np.uint64(2 ** 40)                                                      # saturates at 4 bytes (quirk B12)
        out = str(tmp_path / ("t.jf%d" % k))
        kat_amd.jf_write_records(out, k, canonical, keys, counts)
        hdr, off, raw = read_header(out)
        assert hdr["format"] == "binary/sorted" and hdr["key_len"] == 2 * k and hdr["counter_len"] == 4 and hdr["val_len"] == 7
        assert hdr["max_reprobe"] == 126 and hdr["canonical"] is canonical and hdr["alignment"] == 8 and off % 8 == 0
        assert list(hdr) == sorted(hdr) and len(hdr["reprobes"]) == 127 and hdr["reprobes"][:5] == [1, 1, 3, 6, 10]
        m = hdr["matrix1"]
        assert m["c"] == 2 * k and len(m["columns"]) == 2 * k and hdr["size"] == 1 << m["r"] and hdr["size"] >= min(2 * n, 4 ** k)
        rec = (2 * k + 7) // 8 + 4
        assert (len(raw) - off) == n * rec
        k2, can2, keys2, counts2 = kat_amd.jf_read_records(out)
        assert (k2, can2) == (k, canonical)
        order = [(matrix_times(m["columns"], int(key)) & (hdr["size"] - 1), int(key)) for key in keys2]
        assert order == sorted(order)
        o1, o2 = np.argsort(keys), np.argsort(keys2)
        assert np.array_equal(keys[o1], keys2[o2]) and np.array_equal(np.minimum(counts[o1], np.uint64(2 ** 32 - 1)), counts2[o2])
        t = ko.Table.from_jf(out)                                                               # the oracle's reader agrees
        assert t.k == k and t.n_records == n and t.total == int(np.minimum(counts, np.uint64(2 ** 32 - 1)).sum())


def test_reader_errors(tmp_path):
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.jf_read_records(str(tmp_path / "missing.jf"))
    assert ei.value.code == 2
    bad = tmp_path / "bad.jf"
    bad.write_bytes(b"this is not a jellyfish hash\n")
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.jf_read_records(str(bad))
    assert ei.value.code == 3 and "Failed to parse header of file" in ei.value.message
    js = b'{"format":"bloomcounter","key_len":54,"counter_len":4}'
    bloom = tmp_path / "bloom.jf"
    bloom.write_bytes(b"%09d" % len(js) + js)
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.jf_read_records(str(bloom))
    assert "does not currently support bloom counted" in ei.value.message
    js = b'{"format":"binary/sorted","key_len":54,"counter_len":4}'
    trunc = tmp_path / "trunc.jf"
    trunc.write_bytes(b"%09d" % len(js) + js + b"\x00" * 12)
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.jf_read_records(str(trunc))
    assert "must be a multiple of the length of a record (11)" in ei.value.message
    big = tmp_path / "k40.jf"
    js = b'{"format":"binary/sorted","key_len":80,"counter_len":4}'
    big.write_bytes(b"%09d" % len(js) + js)
    with pytest.raises(kat_amd.KatGpuError) as ei:
        kat_amd.jf_read_records(str(big))
    assert ei.value.code == 6


def test_header_strings_are_escaped(tmp_path, monkeypatch):
    """The reference writes its header through jsoncpp, which escapes; a working directory with a quote or a backslash in its
    name must still give a header every JSON reader (jsoncpp in jellyfish / KAT, json here, our own) accepts."""
    d = tmp_path / 'odd "dir" \\ name'
    d.mkdir()
    monkeypatch.chdir(d)
    keys = np.array([3, 9, 77], np.uint64)
    counts = np.array([1, 2, 3], np.uint64)
    kat_amd.jf_write_records("t.jf", 21, True, keys, counts)
    hdr, _, _ = read_header("t.jf")                                   # json.loads: a strict parser
    assert hdr["pwd"] == str(d) and hdr["format"] == "binary/sorted"
    k, canonical, rk, rc = kat_amd.jf_read_records("t.jf")
    o = np.argsort(rk)
    assert (k, canonical) == (21, True) and np.array_equal(rk[o], keys) and np.array_equal(rc[o], counts)
