#!/usr/bin/env python3
"""Tabulates the pipelines a time-out dumped (option STALL_DUMP_FILE of libg2g.so: one line per strip of the first time-out's DP --
strip, progress word gen:col, HW_ID, queue-pop marker, past-left-chain, past-first-look, the two waves' last publish, heartbeats).
For every file: the HEADS (unfinished strips whose predecessor is finished or >= 48 columns ahead) with the workgroup that holds them
and the wave slot / CU / SE of its first wave, and what the workgroups of a given id range hold.
    python3 tools/stall_tabulate.py [--from 560] file ..."""
import sys


def hwid(x):
    return dict(wave=x & 15, simd=(x >> 4) & 3, pipe=(x >> 6) & 3, cu=(x >> 8) & 15, sh=(x >> 12) & 1, se=(x >> 13) & 7, tg=(x >> 16) & 15)


def main():
    args = sys.argv[1:]
    lo = 560
    if args and args[0] == "--from":
        lo = int(args[1]); args = args[2:]
    heads_total, held, stuck = 0, {}, {}
    for fn in args:
        rows = [l.split() for l in open(fn) if l.strip() and not l.startswith("#")]
        col = {}
        for r in rows:
            g, c = r[1].split(":")
            col[int(r[0])] = int(c) if g != "0" else None
        print("== %s: %d strips" % (fn, len(rows)))
        for r in rows:
            k, c = int(r[0]), col[int(r[0])]
            if c == 1048575:
                fin = True
            else:
                fin = False
            cp = col.get(k - 1, 1048575)
            head = (not fin) and (cp == 1048575 or (cp is not None and (c or 0) + 48 <= cp))
            wg = None if r[3] == "7fffffff" else int(r[3], 16) & 0xFFFF
            if wg is not None:
                held[wg] = held.get(wg, 0) + 1
                if head:
                    stuck[wg] = stuck.get(wg, 0) + 1
            if head:
                heads_total += 1
                h = "" if r[2] == "7fffffff" else " first wave: %s" % hwid(int(r[2], 16))
                print("   head: strip %d at col %s (predecessor %s), workgroup %s%s" % (k, c, "finished" if cp == 1048575 else cp, wg, h))
            elif wg is not None and wg >= lo:
                print("   strip %d (%s) held by workgroup %d: %s" % (k, "finished" if fin else "col %s" % c, wg, "tight behind its predecessor"))
    print("heads: %d; by workgroup id: %s" % (heads_total, " ".join("%d:%d/%d" % (w, stuck.get(w, 0), held[w]) for w in sorted(held) if w >= lo)))


if __name__ == "__main__":
    main()
