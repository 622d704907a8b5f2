"""On-disk formats of the reference that carry an MSA -- SURVEY.md §8 row f4 -- read and written natively so that inputs and
results can be exchanged with `aln` / `prrn5` without the reference in the loop:

  * FASTA (one `>name` header per member);
  * the SEQUENTIAL multi-sequence format (reference sample/pas/Multi_A; Seq::fgetseq, src/seq.h:689-...): a header line
    `<members> <columns>\\t<title>`, then per member `>name`, its aligned row in lines of 60, and a line `/`;
  * the NATIVE interleaved format `prrn5` / `aln` print (reference sample/pas/native_A; src/sqpr.cc): `>title[N] ( l - r )`,
    optional `%` weight lines and `;` annotation lines, then blocks of `<position> <60 columns>| <member>` lines, each block
    closed by a consensus line.

Readers return (names, rows) with rows as equal-length strings over residues and '-'; `read_msa` picks the format from the
first lines.  Writers produce files the reference's own reader accepts (tests/test_seqio.py reads them back through
oracle/_ref).  The per-member gap-run file of IterMsa::prntgap (src/mgaps.cc:31,91-149) is not covered."""
from __future__ import annotations

import re
from typing import List, Sequence, Tuple

_BLOCK = re.compile(r"^\s*\d+\s(.{1,60}?)\s*\|\s*(\S+)\s*$")
_BLOCK_FIXED = re.compile(r"^\s*\d+ {1,2}(.{60})\| (\S+)\s*$")


def _pad(rows: List[str]) -> List[str]:
    width = max((len(r) for r in rows), default=0)
    return [r.ljust(width, "-") for r in rows]


def read_fasta(text: str) -> Tuple[List[str], List[str]]:
    names, rows = [], []
    for line in text.splitlines():
        line = line.rstrip()
        if not line or line.startswith(";"):
            continue
        if line.startswith(">"):
            names.append(line[1:].split()[0] if line[1:].split() else "seq%d" % len(names))
            rows.append("")
        elif rows:
            rows[-1] += "".join(line.split())
    return names, rows


def read_sequential(text: str) -> Tuple[List[str], List[str]]:
    lines = text.splitlines()
    head = lines[0].split()
    n = int(head[0])
    names, rows = [], []
    for line in lines[1:]:
        s = line.rstrip()
        if s.startswith(">"):
            names.append(s[1:].split()[0])
            rows.append("")
        elif s.strip() == "/":
            continue
        elif rows and s and not s.startswith(";") and not s.startswith("%"):
            rows[-1] += "".join(s.split())
    if len(names) != n:
        raise ValueError("sequential MSA: header says %d members, found %d" % (n, len(names)))
    return names, _pad(rows)


def read_native(text: str) -> Tuple[List[str], List[str]]:
    order: List[str] = []
    rows = {}
    for line in text.splitlines():
        m = _BLOCK_FIXED.match(line) or _BLOCK.match(line)
        if not m:
            continue
        seg, name = m.group(1), m.group(2)
        if name not in rows:
            rows[name] = ""
            order.append(name)
        rows[name] += seg.replace(" ", "-")
    out = [rows[k].rstrip() for k in order]
    return order, _pad([r.replace(" ", "-") for r in out])


def sniff(text: str) -> str:
    lines = [l for l in text.splitlines() if l.strip()]
    if not lines:
        raise ValueError("empty file")
    if re.match(r"^\s*\d+\s+\d+(\s|$)", lines[0]):
        return "sequential"
    if any(_BLOCK.match(l) for l in lines[:40]):
        return "native"
    if lines[0].startswith(">"):
        return "fasta"
    raise ValueError("unknown MSA format")


def read_msa(path: str) -> Tuple[List[str], List[str]]:
    text = open(path).read()
    kind = sniff(text)
    names, rows = {"fasta": read_fasta, "sequential": read_sequential, "native": read_native}[kind](text)
    if len({len(r) for r in rows}) > 1 and kind != "fasta":
        raise ValueError("rows of unequal length in %s" % path)
    return names, rows


def write_fasta(path: str, names: Sequence[str], rows: Sequence[str], width: int = 60) -> None:
    with open(path, "w") as fd:
        for nm, r in zip(names, rows):
            fd.write(">%s\n" % nm)
            for i in range(0, len(r), width):
                fd.write(r[i:i + width] + "\n")


def write_sequential(path: str, names: Sequence[str], rows: Sequence[str], title: str = "msa") -> None:
    with open(path, "w") as fd:
        fd.write("%5d %5d\t%s\n" % (len(rows), len(rows[0]) if rows else 0, title))
        for nm, r in zip(names, rows):
            fd.write(">%s\n" % nm)
            for i in range(0, len(r), 60):
                fd.write(r[i:i + 60] + "\n")
            fd.write("/\n")


def write_native(path: str, names: Sequence[str], rows: Sequence[str], title: str = "msa") -> None:
    """Blocks of 60 columns; the position printed in front of a row is the 1-based index of the member's next residue,
    as the reference prints it.  The consensus line is left blank (the reference's reader skips it)."""
    n, length = len(rows), len(rows[0]) if rows else 0
    with open(path, "w") as fd:
        fd.write(">%s[%d] ( 1 - %d )\n\n" % (title, n, length))
        seen = [0] * n
        for c0 in range(0, length, 60):
            for i, (nm, r) in enumerate(zip(names, rows)):
                seg = r[c0:c0 + 60]
                fd.write("%6d  %-60s| %s\n" % (seen[i] + 1, seg, nm))
                seen[i] += sum(1 for ch in seg if ch != "-")
            fd.write("\n\n")
