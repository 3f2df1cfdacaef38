#!/usr/bin/env python3
"""PROFILING_DATUM summariser (the job of the reference's scripts/experiments/analyse.sh:16-43,
without the `q` SQL-on-CSV tool).

  scripts/analyse.py RESULTFILE [...]        (plain text or .gz)

For every result file `…result…` it writes, with the reference's naming (`result` replaced):
  …profiling_data…            the extracted rows  method,context,milliseconds,language
  …profile_summary…           method,context,language,calls,minimum,mean,maximum,total  ordered by total desc
  …profile_summary_readable…  the same, column-aligned
and prints the readable summary.
"""
import collections
import gzip
import os
import re
import sys

ROW = re.compile(r'^PROFILING_DATUM\(\s*"([^"]*)"\s*,\s*"([^"]*)"\s*,\s*([-+0-9.eE]+)\s*,\s*"([^"]*)"\s*\)')


def read_lines(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt", errors="replace") as f:
        yield from f


def extract(path):
    """(method, context, ms, language) of every well-formed datum line; lines carrying debug info
    (DINFO) are dropped as the reference's sed does."""
    rows = []
    for line in read_lines(path):
        if "DINFO" in line:
            continue
        m = ROW.match(line.strip())
        if m:
            rows.append((m.group(1), m.group(2), float(m.group(3)), m.group(4)))
    return rows


def summarise(rows):
    acc = collections.OrderedDict()
    for method, ctx, ms, lang in rows:
        acc.setdefault((method, ctx, lang), []).append(ms)
    out = [(k[0], k[1], k[2], len(v), min(v), sum(v) / len(v), max(v), sum(v)) for k, v in acc.items()]
    out.sort(key=lambda r: -r[7])
    return out


def out_name(path, what):
    """`result` -> `what` in the FILE name (the reference's sed rewrites the whole path, which breaks
    on its own results-<id>/ folders)."""
    base = path[:-3] if path.endswith(".gz") else path
    d, f = os.path.split(base)
    return os.path.join(d, f.replace("result", what) if "result" in f else f + "." + what)


def main(paths):
    if not paths:
        sys.exit(__doc__)
    for path in paths:
        rows = extract(path)
        with open(out_name(path, "profiling_data"), "w") as f:
            for r in rows:
                f.write(f"{r[0]},{r[1]},{r[2]:.6g},{r[3]}\n")
        summ = summarise(rows)
        header = ("method", "context", "language", "calls", "minimum", "mean", "maximum", "total")
        table = [header] + [(a, b, c, str(n), f"{mn:.6g}", f"{me:.6g}", f"{mx:.6g}", f"{to:.6g}")
                            for a, b, c, n, mn, me, mx, to in summ]
        with open(out_name(path, "profile_summary"), "w") as f:
            for r in table:
                f.write(",".join(r) + "\n")
        widths = [max(len(r[i]) for r in table) for i in range(len(header))]
        readable = "\n".join("  ".join(c.ljust(w) for c, w in zip(r, widths)).rstrip() for r in table) + "\n"
        with open(out_name(path, "profile_summary_readable"), "w") as f:
            f.write(readable)
        print(f"== {path}: {len(rows)} data points\n{readable}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
