"""An independent, deliberately naive pure-Python statement of SURVEY.md Appendix D (count semantics) and of the
FASTA/FASTQ record rules, used to cross-check the C oracle and the product's ingest parser on small inputs."""
import gzip
from collections import Counter

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def count_string(seq, k, canonical, counts=None):
    """Every k-window of every maximal ACGTacgt run; canonical = min(w, revcomp(w)) with A<C<G<T."""
    counts = Counter() if counts is None else counts
    run = []
    for ch in seq + "N":
        u = ch.upper() if ch in "ACGTacgt" else None
        if u:
            run.append(u)
            continue
        r = "".join(run)
        for i in range(len(r) - k + 1):
            w = r[i:i + k]
            if canonical:
                w = min(w, revcomp(w))
            counts[w] += 1
        run = []
    return counts


def records(path):
    """Sequence strings of a FASTA/FASTQ(.gz) file: header lines dropped, lines joined, FASTQ qualities skipped by length."""
    data = (gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")).read().decode("latin-1")
    lines = data.split("\n")
    if data.endswith("\n"):
        lines.pop()
    out = []
    if not data:
        return out
    if data[0] == ">":
        cur = None
        for ln in lines:
            if ln.startswith(">"):
                if cur is not None:
                    out.append(cur)
                cur = ""
            elif ln != "":
                cur += ln
        if cur is not None:
            out.append(cur)
    elif data[0] == "@":
        i = 0
        while i < len(lines):
            assert lines[i].startswith("@"), (i, lines[i])
            i += 1
            seq = ""
            while i < len(lines) and not lines[i].startswith("+"):
                seq += lines[i]
                i += 1
            i += 1                       # '+' line
            q = 0
            while i < len(lines) and q < len(seq):
                q += len(lines[i])
                i += 1
            assert q == len(seq), "bad quality length"
            out.append(seq)
    else:
        raise ValueError("Unsupported format")
    return out


def pack(kmer):
    v = 0
    for ch in kmer:
        v = (v << 2) | "ACGT".index(ch)
    return v


def count_files(paths, k, canonical):
    c = Counter()
    for p in paths:
        for r in records(p):
            count_string(r, k, canonical, c)
    return c
