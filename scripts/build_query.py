#!/usr/bin/env python3
"""Collect the apps' SQL INSERT rows from a results folder (the job of the reference's
scripts/experiments/postprocessing/build_query.sh:24 and build_query_from_archives.sh),
optionally as CSV.

  scripts/build_query.py RESULTS_DIR TABLE [--out query.sql] [--csv rows.csv]

Walks RESULTS_DIR recursively (plain result files, .gz, and .tar.gz archives of them), keeps every
`INSERT INTO table_name … VALUES …;` statement with `table_name` replaced by TABLE, and writes them
to query.sql.  --csv additionally flattens the value tuples into one CSV row each with the header
time,correct,kernel,global,local,host,device,matrix,iteration,trial,statistic,experiment_id
(inc/sql_stat.h:23-30 of the reference; same columns here).
"""
import argparse
import csv
import gzip
import io
import os
import sys
import tarfile

COLUMNS = ["time", "correct", "kernel", "global", "local", "host", "device", "matrix", "iteration", "trial",
           "statistic", "experiment_id"]


def texts(root):
    for d, _, files in os.walk(root):
        for fn in sorted(files):
            p = os.path.join(d, fn)
            try:
                if fn.endswith((".tar.gz", ".tgz")):
                    with tarfile.open(p, "r:gz") as t:
                        for m in t.getmembers():
                            if m.isfile():
                                yield p + ":" + m.name, t.extractfile(m).read().decode(errors="replace")
                elif fn.endswith(".gz"):
                    yield p, gzip.open(p, "rt", errors="replace").read()
                elif fn.endswith((".txt", ".log", ".out")):
                    yield p, open(p, errors="replace").read()
            except (OSError, tarfile.TarError, EOFError) as e:
                print(f"skipping {p}: {e}", file=sys.stderr)


def statements(root):
    for src, text in texts(root):
        for line in text.splitlines():
            i = line.find("INSERT")
            if i >= 0:
                yield src, line[i:].strip()


def tuples(values):
    """Top-level (...) groups of a VALUES list; quotes protect parentheses and commas
    (device names look like "AMD Instinct MI355X (gfx950)")."""
    start, depth, quoted = None, 0, False
    for i, ch in enumerate(values):
        if ch == '"':
            quoted = not quoted
        elif quoted:
            continue
        elif ch == "(":
            if depth == 0:
                start = i + 1
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0 and start is not None:
                yield values[start:i]
                start = None


def rows_of(stmt):
    values = stmt.split("VALUES", 1)[1] if "VALUES" in stmt else ""
    for body in tuples(values):
        fields = next(csv.reader(io.StringIO(body), skipinitialspace=True))
        if len(fields) == len(COLUMNS):
            yield [f.strip() for f in fields]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("results_dir"), ap.add_argument("table")
    ap.add_argument("--out", default="query.sql")
    ap.add_argument("--csv", default=None)
    a = ap.parse_args(argv)
    n_stmt = n_rows = 0
    cw = None
    if a.csv:
        cf = open(a.csv, "w", newline="")
        cw = csv.writer(cf)
        cw.writerow(COLUMNS)
    with open(a.out, "w") as q:
        for _, stmt in statements(a.results_dir):
            q.write(stmt.replace("table_name", a.table) + "\n")
            n_stmt += 1
            if cw:
                for r in rows_of(stmt):
                    cw.writerow(r)
                    n_rows += 1
    print(f"{n_stmt} INSERT statements -> {a.out}" + (f", {n_rows} rows -> {a.csv}" if a.csv else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
