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


# ---- kat sect (src/sect.cc), stated independently of oracle/koracle_sect.c ----

def seqan_records(path):
    """(name, seq) byte strings the way SeqAn 2.0.0's SeqFileIn hands them to Sect: the name is the header line, the
    sequence is everything up to the next '>' ('+' for FASTQ) minus CR/LF; FASTQ qualities are skipped by count."""
    raw = (gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")).read()
    base = str(path)[:-3] if str(path).lower().endswith(".gz") else str(path)
    low = base.lower()
    if low.endswith((".fa", ".fasta")):
        fastq = False
    elif low.endswith((".fq", ".fastq")):
        fastq = True
    elif low.endswith(".txt"):                     # SeqAn's "Raw" format: every line is a record without a name
        out, i, n = [], 0, len(raw)
        while i < n:
            j = i
            while j < n and raw[j:j + 1] not in (b"\n", b"\r"):
                j += 1
            out.append((b"", raw[i:j]))
            if raw[j:j + 1] == b"\r":
                j += 1
            if raw[j:j + 1] == b"\n":
                j += 1
            i = j
        return out
    else:                                          # SeqAn decides on the file name alone
        raise ValueError("Unknown file extension of %s: iostream error" % path)
    begin, stop = (b"@", b"+") if fastq else (b">", b">")
    out = []
    i, n = 0, len(raw)

    def eat_line(i):
        j = i
        while j < n and raw[j:j + 1] not in (b"\n", b"\r"):
            j += 1
        line = raw[i:j]
        if raw[j:j + 1] == b"\r":
            j += 1
        if raw[j:j + 1] == b"\n":
            j += 1
        return line, j

    while i < n:
        j = raw.find(begin, i)
        if j < 0:
            raise ValueError("Unexpected end of input.")
        name, i = eat_line(j + 1)
        j = raw.find(stop, i)
        if j < 0:
            if fastq:
                raise ValueError("Unexpected end of input.")
            j = n
        seq = raw[i:j].replace(b"\n", b"").replace(b"\r", b"")
        i = j
        if fastq:
            _, i = eat_line(i + 1)
            left = len(seq)
            while i < n and left:
                if raw[i:i + 1] not in (b"\n", b"\r"):
                    left -= 1
                i += 1
            j = raw.find(b"@", i)
            i = n if j < 0 else j
        out.append((name, seq))
    return out


def _u32(x):
    return x & 0xFFFFFFFF


def _f5(x):
    if x != x:
        return "-nan"           # 0.0/0.0 at run time on x86-64 is the negative default NaN; glibc prints "-nan"
    return "%.5f" % x


def sect(counts, k, canonical, seq_path, gc_bins=1001, cvg_bins=1001, output_gc_stats=False, no_count_stats=False,
         extract_nr=False, extract_r=False, min_repeat=2, max_repeat=0, save=False):
    """`counts` maps upper-case k-mer strings (canonical ones if the hash is canonical) to counts.  Returns
    {file suffix: bytes} for every file `kat sect` would write."""
    files = {}
    cvg, gcf, nr, rr = [], [], [], []
    stats = [b"seq_name\tmedian\tmean\tgc%\tseq_length\tkmers_in_seq\tinvalid_kmers\t%_invalid\tnon_zero_kmers\t%_non_zero\t%_non_zero_corrected\n"]
    mx = [[0] * cvg_bins for _ in range(gc_bins)]

    def regions(sink, name, seq, cs, lo, hi):
        maxs = ("-%d" % hi) if hi > 0 else "+"
        idx, start, inside, acc = 1, 0, False, b""

        def emit(end, tail):
            nonlocal idx
            sink.append(b">" + name + ("___region:%d_length:%d_pos:%d:%d_cov:%d%s\n" % (idx, _u32(end - start - 1), start + 1, end, lo, maxs)).encode()
                        + acc + tail + b"\n")
            idx += 1
        for j, c in enumerate(cs):
            if c >= lo and (c <= hi or hi == 0):
                if not inside:
                    start, inside = j, True
                acc += seq[j:j + 1]
            elif inside:
                end = j + k - 1
                emit(end, seq[j + 1:end])
                inside, acc = False, b""
        if inside:
            end = len(cs) + k - 1
            emit(end, seq[len(cs):end])

    for name, seq in seqan_records(seq_path):
        L = len(seq)
        nb = L - k + 1
        cs, gs = [], []
        s = seq.decode("latin-1")
        for i in range(max(nb, 0)):
            w = s[i:i + k]
            if any(ch not in "ACGTacgt" for ch in w):
                cs.append(0)
                gs.append(-1)
                continue
            u = w.upper()
            if canonical:
                u = min(u, revcomp(u))
            cs.append(counts.get(u, 0))
            gs.append(sum(ch in "GCgc" for ch in w))
        invalid = sum(g < 0 for g in gs)
        nonzero = sum(1 for c, g in zip(cs, gs) if g >= 0 and c)
        median = _u32(sorted(cs)[len(cs) // 2]) if cs else 0
        mean = (sum(cs) / nb) if cs else 0.0
        p_nz = 0.0 if nonzero == 0 or nb <= 0 else nonzero / nb * 100.0
        p_inv = 0.0 if invalid == 0 or nb <= 0 else invalid / nb * 100.0
        not_invalid = (nb - invalid) & 0xFFFFFFFFFFFFFFFF
        p_nzc = 0.0 if nonzero == 0 or not_invalid == 0 else nonzero / not_invalid * 100.0
        g_ = sum(ch in "Gg" for ch in s)
        c_ = sum(ch in "Cc" for ch in s)
        n_ = sum(ch in "Nn" for ch in s)
        gc = (g_ + c_) / (L - n_) if L - n_ else float("nan")
        x = 0 if gc != gc else int(gc * gc_bins) & 0xFFFF
        if x < gc_bins and cvg_bins > 0:
            mx[x][0] += L                      # average_cvg is never assigned in the reference: always the first coverage bin
        if not no_count_stats:
            cvg.append(b">" + name + b"\n" + (" ".join(map(str, cs)) if cs else "0").encode() + b"\n")
        if output_gc_stats:
            gcf.append(b">" + name + b"\n" + (" ".join("-0.1" if g < 0 else "%.1f" % (g / k * 100.0) for g in gs) if gs else "0.0").encode() + b"\n")
        if extract_nr:
            regions(nr, name, seq, cs, 1, min_repeat)
        if extract_r:
            regions(rr, name, seq, cs, min_repeat, max_repeat)
        stats.append(name + ("\t%d\t%s\t%s\t%d\t%d\t%d\t%s\t%d\t%s\t%s\n" % (median, _f5(mean), _f5(gc), _u32(L), _u32(_u32(L) - k + 1), _u32(invalid),
                                                                          _f5(p_inv), _u32(nonzero), _f5(p_nz), _f5(p_nzc))).encode())
    if not no_count_stats:
        files["-counts.cvg"] = b"".join(cvg)
    if output_gc_stats:
        files["-counts.gc"] = b"".join(gcf)
    if extract_nr:
        files["-non_repetitive.fa"] = b"".join(nr)
    if extract_r:
        files["-repetitive.fa"] = b"".join(rr)
    files["-stats.tsv"] = b"".join(stats)
    if save:
        head = "# Title:Contamination Plot for %s and \"\"\n# XLabel:GC%%\n# YLabel:Average K-mer Coverage\n# ZLabel:Base Count per bin\n" % seq_path
        head += "# Columns:%d\n# Rows:%d\n# MaxVal:%d\n# Transpose:0\n###\n" % (gc_bins, cvg_bins, max(max(r) for r in mx))
        files["-contamination.mx"] = (head + "".join(" ".join(map(str, r)) + "\n" for r in mx)).encode()
    return files


def cold(reads, canon_reads, asm, canon_asm, k, asm_path):
    """`kat cold` (src/cold.cc): the bytes of <prefix>-stats.tsv.  reads / asm map upper-case k-mers to counts."""
    out = [b"seq_name\tread_median_cvg\tread_mean_cvg\tasm_cn\tgc%\tseq_length\tkmers_in_seq\tinvalid_kmers\t%_invalid\tnon_zero_kmers\t%_non_zero\t%_non_zero_corrected\n"]
    for name, seq in seqan_records(asm_path):
        s = seq.decode("latin-1")
        L = len(s)
        nb = L - k + 1
        rs, as_, invalid = [], [], 0
        for i in range(max(nb, 0)):
            w = s[i:i + k]
            if any(ch not in "ACGTacgt" for ch in w):
                rs.append(0)
                as_.append(0)
                invalid += 1
                continue
            u = w.upper()
            rs.append(reads.get(min(u, revcomp(u)) if canon_reads else u, 0))
            as_.append(asm.get(min(u, revcomp(u)) if canon_asm else u, 0))
        nonzero = sum(1 for c in rs if c)
        median = _u32(sorted(rs)[len(rs) // 2]) if rs else 0
        asm_cn = _u32(sorted(as_)[len(as_) // 2]) if as_ else 0
        mean = sum(rs) / nb if rs else 0.0
        p_nz = 0.0 if nonzero == 0 or nb <= 0 else nonzero / nb * 100.0
        p_inv = 0.0 if invalid == 0 or nb <= 0 else invalid / nb * 100.0
        not_invalid = (nb - invalid) & 0xFFFFFFFFFFFFFFFF
        p_nzc = 0.0 if nonzero == 0 or not_invalid == 0 else nonzero / not_invalid * 100.0
        g_ = sum(ch in "GgCc" for ch in s)
        n_ = sum(ch in "Nn" for ch in s)
        gc = g_ / (L - n_) if L - n_ else float("nan")
        out.append(name + ("\t%d\t%s\t%d\t%s\t%d\t%d\t%d\t%s\t%d\t%s\t%s\n" % (median, _f5(mean), asm_cn, _f5(gc), _u32(L), _u32(_u32(L) - k + 1),
                                                                                 _u32(invalid), _f5(p_inv), _u32(nonzero), _f5(p_nz), _f5(p_nzc))).encode())
    return b"".join(out)
