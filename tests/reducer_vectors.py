"""Shared by tests/golden/make_reducer_vectors.py (which writes the golden file from the reference's own code) and the tests that hold the
oracle and the HIP path to it: the cases, the way the reference's pieces are driven, and the text form the matrices are compared in."""
import hashlib
import math
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JF_REF = os.path.join(ROOT, "oracle", "_ref", "jf_ref")
KAT_REF = os.path.join(ROOT, "oracle", "_ref", "kat_ref_parts")
GOLDEN = os.path.join(ROOT, "tests", "golden", "reducer_vectors.json")
R1, R2 = "ecoli_r1.1K.fastq", "ecoli_r2.1K.fastq"

# (k, d1_scale, d2_scale, d1_bins, d2_bins): KAT's defaults, scaled axes, folded (few bins: the catch-all cells), unequal bins
COMP_CASES = [(13, 1.0, 1.0, 1001, 1001), (21, 1.0, 1.0, 1001, 1001), (27, 1.0, 1.0, 1001, 1001), (13, 0.5, 0.25, 1001, 1001), (13, 1.0, 1.0, 6, 4),
              (21, 2.0, 0.1, 12, 30), (27, 3.0, 1.0, 5, 5)]
# (k, cvg_scale, cvg_bins): defaults, scaled, folded
GCP_CASES = [(13, 1.0, 1000), (21, 1.0, 1000), (27, 1.0, 1000), (13, 0.1, 1000), (17, 1.0, 3), (27, 2.5, 7)]


def tag(case):
    return ":".join(str(x) for x in case)


def digest(b):
    return hashlib.sha256(b if isinstance(b, bytes) else b.encode()).hexdigest()


def run(binary, args, stdin=None):
    r = subprocess.run([binary] + [str(a) for a in args], input=stdin, capture_output=True, timeout=600)
    assert r.returncode == 0, (binary, args, r.returncode, r.stderr[:200])
    return r.stdout


def ref_counts(path, k):
    """{k-mer string: count} as the reference's parser + mer_iterator deliver them (canonical)."""
    out = run(JF_REF, ["kmers", k, 1, path])
    return {a.decode(): int(b) for a, b in (l.split() for l in out.splitlines())}


def scale_counter(c, f):                                   # src/comp.hpp:303-306
    return 0 if c == 0 else int(math.ceil(float(c) * f))


def matrix_text(rows):
    """rows: iterable of rows of ints -> the text SparseMatrix::printMatrix writes (values separated by one blank, one row per line)."""
    return "".join(" ".join(str(int(v)) for v in r) + "\n" for r in rows)


def reference_comp(refdata, k, s1, s2, b1, b2):
    h1, h2 = ref_counts(os.path.join(refdata, R1), k), ref_counts(os.path.join(refdata, R2), k)
    n = min(b1, b2)
    lines = ["%d" % n]
    cells = {}

    def inc(i, j):
        cells[(i, j)] = cells.get((i, j), 0) + 1
    for key, c1 in h1.items():                              # src/comp.cc:393-425: every k-mer of hash 1
        c2 = h2.get(key, 0)
        lines.append("h1 %d %d" % (c1, c2))
        inc(min(scale_counter(c1, s1), b1 - 1), min(scale_counter(c2, s2), b2 - 1))
    for key, c2 in h2.items():                              # src/comp.cc:441-473: every k-mer of hash 2; the matrix only for those hash 1 lacks
        c1 = h1.get(key, 0)
        lines.append("h2 %d %d" % (c1, c2))
        if c1 == 0:
            inc(0, min(scale_counter(c2, s2), b2 - 1))
    out = run(KAT_REF, ["compupdate"], ("\n".join(lines) + "\n").encode()).decode().splitlines()
    counters = [int(x) for x in out[0].split()]
    spectra = "\n".join(out[1:5]) + "\n"
    mx = run(KAT_REF, ["matrix", b1, b2], "".join("%d %d %d\n" % (i, j, v) for (i, j), v in sorted(cells.items())).encode()).decode()
    maxval, body = mx.split("\n", 1)
    return counters, spectra, int(maxval), body


def reference_gcp(refdata, k, scale, bins):
    h = ref_counts(os.path.join(refdata, R1), k)
    keys = list(h)
    gc = [int(l.split()[1]) for l in run(KAT_REF, ["strutils"], ("\n".join(keys) + "\n").encode()).decode().splitlines()]      # the reference's gcCount
    cells = {}
    for key, g in zip(keys, gc):
        c = h[key]
        pos = 0 if c == 0 else int(math.ceil(float(c) * scale))                                                              # src/gcp.cc:188-195
        pos = bins if pos > bins else pos
        cells[(g, pos)] = cells.get((g, pos), 0) + 1
    # Gcp's matrix has k rows (src/gcp.cc:93): a k-mer of GC count k lies outside and SparseMatrix::inc drops it (quirk B1) -- the real class decides
    mx = run(KAT_REF, ["matrix", k, bins + 1], "".join("%d %d %d\n" % (i, j, v) for (i, j), v in sorted(cells.items())).encode()).decode()
    maxval, body = mx.split("\n", 1)
    return int(maxval), body
